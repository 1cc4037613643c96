"""Parity at the widths the benchmark runs (BASELINE.json configs[1]: 6 blocks, base 100, growth 1.5 ->
100 / 150 / 225 / 337 / 506 / 759 channels, 128 x 10 s @ 44.1 kHz, mel_2048_1024_128).

* every convolution layer of that model through fwd / dgrad / accumulating dgrad / wgrad, in the default split-fp16
  arithmetic AND in native fp32, against PyTorch's fp64 convolution on the CPU (never against another HIP kernel),
  with the kernel instantiation asserted to be the one batch 128 selects;
* one full training forward / backward of the real 21.5 M-parameter model against the fixture the imported reference
  produced (tests/golden/g12_cfg2_step.npz) and against the CPU oracle in fp32 and fp64;
* the split-fp16 arithmetic where it can break: per-sample gradient rows spread over 2^-40, Inf / NaN operands, an
  LSEP overflow step.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import torch.nn.functional as TF  # noqa: E402

from freesound_classification_amd import functional as F  # noqa: E402
from freesound_classification_amd.networks.classifiers import TwoDimensionalCNNClassificationModel  # noqa: E402
from freesound_classification_amd.networks.losses import lsep_loss  # noqa: E402
from oracle import ref_torch as oref  # noqa: E402
from test_oracle_cpu import cfg2_golden_inputs, check_cfg2_step_against_golden  # noqa: E402

DEV = torch.device("cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, "gpurun_out", "cfg2_layer_parity.txt")


def cfg2_layers():
    """(c_in, c_out, h, w, k) of the 24 convolutions of the cfg-2 model (classifiers.py:524-536, 72-104)."""
    out = []
    h, w, c_in = 128, 431, 2
    for depth in [int(1.5 ** k * 100) for k in range(6)]:
        out.append((c_in, depth, h, w, 3))
        h, w = h // 2, w // 2
        out += [(depth, depth, h, w, 1), (depth, depth, h, w, 3)]        # conv1 == conv3 (1x1), conv2 (3x3)
        c_in = depth
    return out


LAYERS = cfg2_layers()


def _batch_for(layer, arith):
    """Batch 128 for the planes up to 8 x 26 (cheap enough in fp64); otherwise the smallest batch whose three kernel
    instantiations (fwd, dgrad, wgrad) are the ones batch 128 selects."""
    c_in, c_out, h, w, k = layer
    if h * w <= 8 * 26:
        return 128
    full = [F.plan_name(F._desc(128, c_in, c_out, h, w, k, k, arith), m) for m in (0, 1, 2)]
    for n in (2, 4, 8, 16, 32, 64):
        if [F.plan_name(F._desc(n, c_in, c_out, h, w, k, k, arith), m) for m in (0, 1, 2)] == full:
            return n
    return 128


def _report(line):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(line + "\n")


# (batch 4 on the planes above 32 x 107 is covered by the batch-128 plan at batch 2 .. 64: not generated)
_LAYER_CASES = [(l, a, b) for l in LAYERS for a in (3, 0) for b in ("bench", 4) if not (b == 4 and l[2] * l[3] > 32 * 107)]


@pytest.mark.parametrize("layer,arith,batch", _LAYER_CASES,
                         ids=["%dto%d_%dx%d_k%d" % l + "-" + {3: "f16x3", 0: "f32"}[a] + "-" + ("n128plan" if b == "bench" else "n4")
                              for l, a, b in _LAYER_CASES])
def test_cfg2_layer_against_fp64(layer, arith, batch):
    """batch "bench": the kernel instantiations (and, up to 8 x 26, the exact plans) batch 128 runs.  batch 4: the
    plans the full-model parity step below runs (split-K, several images per box, partly filled tiles)."""
    c_in, c_out, h, w, k = layer
    n = _batch_for(layer, arith) if batch == "bench" else 4
    names = [F.plan_name(F._desc(n, c_in, c_out, h, w, k, k, arith), m) for m in (0, 1, 2)]
    if batch == "bench":
        assert names == [F.plan_name(F._desc(128, c_in, c_out, h, w, k, k, arith), m) for m in (0, 1, 2)]
        if arith == 3 and c_in >= 32:
            assert names[0].startswith("conv_fwd_x3_kernel") and names[0].endswith(",3>"), names
    torch.manual_seed(c_in * 7 + c_out + h)
    pad = k // 2
    x = torch.randn(n, c_in, h, w)
    wt = torch.randn(c_out, c_in, k, k) / (c_in * k * k) ** 0.5
    b = torch.randn(c_out)
    gy = torch.randn(n, c_out, h, w)
    y64 = TF.conv2d(x.double(), wt.double(), b.double(), padding=pad)
    dx64 = torch.nn.grad.conv2d_input(x.shape, wt.double(), gy.double(), padding=pad)
    dw64 = torch.nn.grad.conv2d_weight(x.double(), wt.shape, gy.double(), padding=pad)
    # what PyTorch's own fp32 convolution loses against fp64 on the same operands: the yard-stick
    e_y = float((TF.conv2d(x, wt, b, padding=pad).double() - y64).abs().max())
    e_dx = float((torch.nn.grad.conv2d_input(x.shape, wt, gy, padding=pad).double() - dx64).abs().max())
    e_dw = float((torch.nn.grad.conv2d_weight(x, wt.shape, gy, padding=pad).double() - dw64).abs().max())

    mode0 = F.get_conv_arith()
    try:
        F.set_conv_arith(arith)
        xd, wd, bd, gd = x.to(DEV), wt.to(DEV), b.to(DEV), gy.to(DEV)
        y = F.conv_forward(xd, wd, bd).cpu()
        dx = F.conv_dgrad(gd, wd, x.shape).cpu()
        base = torch.randn_like(x)
        dxa = F.conv_dgrad(gd, wd, x.shape, accumulate_into=base.to(DEV)).cpu()
        dw = F.conv_wgrad(xd, gd, wt.shape).cpu()
    finally:
        F.set_conv_arith(mode0)
    g_y = float((y.double() - y64).abs().max())
    g_dx = float((dx.double() - dx64).abs().max())
    g_dxa = float((dxa.double() - (dx64 + base.double())).abs().max())
    g_dw = float((dw.double() - dw64).abs().max())
    _report("%-22s arith %d n %3d  fwd %.2e (torch f32 %.2e, x%.2f) %s | dgrad %.2e (%.2e, x%.2f) %s | wgrad %.2e (%.2e, x%.2f) %s"
            % ("%dto%d_%dx%d_k%d" % layer, arith, n, g_y, e_y, g_y / e_y, names[0].split(" ")[0], g_dx, e_dx, g_dx / e_dx,
               names[1].split(" ")[0], g_dw, e_dw, g_dw / e_dw, names[2].split(" ")[0]))
    # Bounds.  (a) 5x the error PyTorch's own (blocked, multi-accumulator) fp32 CPU convolution makes against fp64
    # on the same operands, or (b) the fp32 rounding model of ONE serial accumulation chain: the MFMA kernels add K / 4
    # (native) or K / 32 (split-fp16) partial products into one fp32 accumulator, error ~ eps * sqrt(chain) * |result|.
    # Measured (profiles/r02_cfg2_layer_parity.txt): split-fp16 0.3 .. 3.6x PyTorch (5.7 .. 8.6x on the dgrads with
    # K = 3033 .. 6831, abs 1.7e-5 on |dx| <= 6); the native fp32 kernels 0.3 .. 9.8x; all >= 50x inside 1e-3.
    eps = 2.0 ** -23

    def bound(e32, k_terms, ref):
        return max(5.0 * e32, 2.0 * eps * (k_terms / 4.0) ** 0.5 * float(ref.abs().max())) + 1e-7

    assert g_y < bound(e_y, c_in * k * k, y64), (g_y, e_y)
    assert g_dx < bound(e_dx, c_out * k * k, dx64), (g_dx, e_dx)
    assert g_dxa < bound(e_dx, c_out * k * k, dx64) + 1e-6, (g_dxa, e_dx)
    assert g_dw < bound(e_dw, n * h * w, dw64), (g_dw, e_dw)
    assert max(g_y, g_dx) < 5e-5                      # and in absolute terms: 20x inside the 1e-3 fp32 bar


# ------------------------------------------------------------------------------ the full cfg-2 model
class NS(dict):
    __getattr__ = dict.__getitem__


def cfg2_experiment(dropout=0.0):
    return NS(config=NS(
        network=NS(num_conv_blocks=6, start_deep_supervision_on=1, conv_base_depth=100, growth_rate=1.5,
                   output_dropout=dropout, aggregation_type="max"),
        data=NS(features="mel_2048_1024_128", _input_dim=128, _n_classes=80),
        train=NS(accumulation_steps=1, optimizer="adam", learning_rate=1e-3, weight_decay=0.0,
                 scheduler="1cycle_0.0001_0.005", switch_off_augmentations_on=1000, _save_every=1000)))


@pytest.fixture(scope="module", params=["f16x3", "bf16x9", "f16x6"])
def cfg2_step(golden, request):
    """One training forward / backward + eval forward of the product at cfg-2 width on the fixture's inputs, in the shipped fast
    arithmetic (f16x3: two fp16 limbs, three products) and in the fp32-exact one (bf16x9: three bf16 limbs, nine products -- the
    arithmetic bench.py's headline runs; f16x6: three scaled fp16 limbs, six products); the tests that use the fixture run in the same mode."""
    g = golden("g12_cfg2_step.npz")
    mode0 = F.get_conv_arith()
    F.set_conv_arith(request.param)
    torch.manual_seed(int(g["seed"]))
    m = TwoDimensionalCNNClassificationModel(cfg2_experiment(), device="cuda:0")
    state = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    signal, labels = cfg2_golden_inputs(g)
    m.train()
    logits = m(signal.to(DEV))["class_logits"]
    per = lsep_loss(logits, labels.to(DEV), average=False)
    F.mean(per).backward()
    grads = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}
    m.eval()
    with torch.no_grad():
        ev = m(signal.to(DEV))["class_logits"].cpu()
    yield dict(g=g, state=state, signal=signal, labels=labels, logits=logits.detach().cpu(), per=per.detach().cpu(),
               grads=grads, eval_logits=ev, model=m, arith=request.param)
    F.set_conv_arith(mode0)


def test_cfg2_model_forward_against_reference_golden(cfg2_step):
    """Same seed -> the reference's initial parameters (checksums); then logits, per-sample LSEP and eval-mode logits of
    the real 21.5 M-parameter model against what the imported reference produced: 1e-3 absolute (measured 5e-5)."""
    s = cfg2_step
    g = s["g"]
    assert sum(v.numel() for v in s["grads"].values()) == int(g["n_params"]) == 21545583
    for k, v in s["state"].items():
        if ("init_sum." + k) in g:             # (the fp64 sum depends on the host's thread count in the last bits)
            assert abs(float(v.double().sum()) - float(g["init_sum." + k])) <= 1e-9 * float(g["init_abs." + k]), k
            assert abs(float(v.double().abs().sum()) - float(g["init_abs." + k])) <= 1e-9 * float(g["init_abs." + k]), k
    d_logits = float(np.abs(s["logits"].numpy() - g["logits"]).max())
    d_loss = float(np.abs(s["per"].numpy() - g["loss"]).max())
    d_eval = float(np.abs(s["eval_logits"].numpy() - g["eval_logits"]).max())
    _report("cfg2 model forward vs reference golden: logits %.2e  loss %.2e  eval logits %.2e" % (d_logits, d_loss, d_eval))
    assert d_logits < 1e-3 and d_loss < 1e-3 and d_eval < 1e-3


def _relative_l2(a, b):
    num = sum(float((a[k].double() - b[k].double()).pow(2).sum()) for k in b)
    den = sum(float(b[k].double().pow(2).sum()) for k in b)
    return (num / den) ** 0.5


def test_cfg2_model_gradient_end_to_end_within_the_cpu_paths_own_sensitivity(cfg2_step):
    """End to end, the training gradient of this model is NOT a smooth function at fp32 resolution: 22 M max-pool
    windows, PReLU kinks and global-max picks sit in front of a head whose two BatchNorm1d layers normalise over the
    4 samples of this batch.  Measured when this test was written: on the CPU, in fp32, a 1e-6 RELATIVE perturbation of
    the waveform moves parameter gradients by 3-4 % of their scale; the fp32 CPU path is 0.9 % from its own fp64
    evaluation; the (exactly matching, see the stage-wise test) head turns the 3e-5 feature differences between two
    correct forwards into a 6 % difference of the feature gradient.  An element-wise 1e-3 bound is therefore
    unattainable for ANY two implementations; what is asserted: the accelerated gradient is no further from the CPU
    oracle's than a few times what the oracle moves under that 1e-6 perturbation (global relative L2 over all 21.5 M
    elements, and per tensor on its own scale).  The tight statements are the per-layer tests above (every conv at
    these widths against fp64) and the stage-wise test below (every block and the head on IDENTICAL inputs)."""
    s = cfg2_step

    def oracle_grads(signal):
        ref = oref.TagCNN2d("mel_2048_1024_128", 6, 100, 1.5, 1, 80)
        ref.load_state_dict(s["state"])
        ref.train()
        logits = ref(signal)["class_logits"]
        oref.lsep(logits, s["labels"], average=False).mean().backward()
        return logits.detach(), {k: p.grad.detach().clone() for k, p in ref.named_parameters()}

    rl, r32 = oracle_grads(s["signal"])
    gen = torch.Generator().manual_seed(1)
    _, rpert = oracle_grads(s["signal"] * (1.0 + 1e-6 * torch.randn(s["signal"].shape, generator=gen)))
    assert float((s["logits"] - rl).abs().max()) < 1e-3
    yard = _relative_l2(rpert, r32)
    ours = _relative_l2(s["grads"], r32)
    worst_y = max(float((rpert[k] - r32[k]).abs().max()) / max(1.0, float(r32[k].abs().max())) for k in r32)
    worst_o = max(float((s["grads"][k] - r32[k]).abs().max()) / max(1.0, float(r32[k].abs().max())) for k in r32)
    cos = sum(float((s["grads"][k].double() * r32[k].double()).sum()) for k in r32) / (
        sum(float(s["grads"][k].double().pow(2).sum()) for k in r32) ** 0.5 * sum(float(r32[k].double().pow(2).sum()) for k in r32) ** 0.5)
    _report("cfg2 model gradient end to end: relative L2 to the CPU oracle %.3e (oracle under a 1e-6 input perturbation %.3e); "
            "worst per-tensor scaled diff %.3e (%.3e); cosine %.6f" % (ours, yard, worst_o, worst_y, cos))
    assert ours < 5.0 * yard + 1e-3, (ours, yard)
    assert worst_o < 5.0 * worst_y + 1e-3, (worst_o, worst_y)
    assert cos > 0.99


@pytest.mark.parametrize("stage", [0, 1, 2, 3, 4, 5, "head"])
def test_cfg2_stagewise_gradients_on_identical_inputs(cfg2_step, stage):
    """Every conv block of the cfg-2 model at its real width and plane (100 ch @ 128 x 431 ... 759 ch @ 4 x 13), and
    the classifier head (1977 features), forward AND backward on inputs and upstream gradients that are bit-identical
    for the accelerated block and the CPU oracle's block (the oracle's own activations of the golden batch): output,
    deep-supervision feature, input gradient and all 22 parameter gradients."""
    s = cfg2_step
    ref = oref.TagCNN2d("mel_2048_1024_128", 6, 100, 1.5, 1, 80)
    ref.load_state_dict(s["state"])
    ref.train()
    m = s["model"]
    m.load_state_dict(s["state"])
    m.train()
    for p in m.parameters():
        p.grad = None                       # (the fixture's full step left its gradients there)
    gen = torch.Generator().manual_seed(17)
    with torch.no_grad():
        x = ref.front_end(s["signal"])
        feats = []
        for k, blk in enumerate(ref.conv_modules):
            if k == stage:
                break
            x = blk(x)
            if k >= 1:
                feats.append(TF.adaptive_max_pool2d(x, 1).flatten(1))
    if stage == 0:
        # The zero-padded tails of the golden batch make 8 % of the log-mel image EXACTLY constant (log 1e-4): the stem convolution is
        # then constant over whole regions and every 2 x 2 pooling window there is a four-way tie that each implementation breaks by
        # the last bits of its own convolution arithmetic -- systematically, not rarely.  The winners decide at which row the frequency
        # channel (a ramp along H) is correlated with the gradient: the first BatchNorm's dgamma moves by 3e-3 of its scale between
        # two correct evaluations (round 5: between two builds whose convolution outputs differ by 1e-6), antisymmetrically in the two
        # input channels.  A 1e-5 dither of the identical input removes the exact ties (padded rows are covered by the golden forward
        # test and the front-end tests).
        x = x + 1e-5 * torch.randn(x.shape, generator=gen)
    if stage == "head":
        f_in = torch.cat(feats, -1)
        fr = f_in.clone().requires_grad_()
        rl = ref.output_transform(fr)
        oref.lsep(rl, s["labels"], average=False).mean().backward()
        fp = f_in.to(DEV).requires_grad_()
        ot = m.output_transform
        z = F.bn_act(fp, ot[0], None, True)
        z = F.linear(z, ot[1].weight, ot[1].bias)
        z = F.bn_act(z, ot[2], ot[3], True)
        ml = F.linear(z, ot[5].weight, ot[5].bias)
        F.mean(lsep_loss(ml, s["labels"].to(DEV), average=False)).backward()
        pairs = [("logits", ml, rl), ("d feats", fp.grad, fr.grad)]
        pairs += [("grad " + k, p.grad, dict(ref.output_transform.named_parameters())[k].grad) for k, p in ot.named_parameters()]
    else:
        blk, mods = ref.conv_modules[stage], m.conv_modules[stage]
        xr = x.clone().requires_grad_(stage > 0)
        out_r = blk(xr)
        feat_r = TF.adaptive_max_pool2d(out_r, 1).flatten(1)
        g_out = 1e-3 * torch.randn(out_r.shape, generator=gen)
        g_feat = 1e-2 * torch.randn(feat_r.shape, generator=gen)
        want_head = stage >= 1
        last = stage == 5
        torch.autograd.backward([out_r] + ([feat_r] if want_head else []), [g_out * (0.0 if last else 1.0)] + ([g_feat] if want_head else []))
        xp = x.to(DEV).requires_grad_(stage > 0)
        out_p, feat_p = F.conv_block(xp, mods, True, want_head, 2)
        if last:
            torch.autograd.backward([feat_p], [g_feat.to(DEV)])            # the last block only feeds its head
        else:
            torch.autograd.backward([out_p] + ([feat_p] if want_head else []), [g_out.to(DEV)] + ([g_feat.to(DEV)] if want_head else []))
        pairs = [("out", out_p, out_r)]
        if want_head:
            pairs.append(("feat", feat_p, feat_r))
        if stage > 0:
            pairs.append(("dx", xp.grad, xr.grad))
        rp = dict(blk.named_parameters())
        pairs += [("grad " + k, p.grad, rp[k].grad) for k, p in mods.named_parameters()]

        def fp64_grads():
            """The same block in fp64 on the same input and upstream gradients: the yard-stick for a parameter gradient whose fp32
            evaluation is ill-conditioned (the first BatchNorm's dgamma / dbeta: cancelling sums over 0.2 ... 5.5 M positions, whose
            fp32 value moves by a few 1e-3 of the scale under ANY change of summation order or of the last bits upstream)."""
            import copy
            b64 = copy.deepcopy(blk).double()
            for q in b64.parameters():
                q.grad = None
            o64 = b64(x.double())
            f64 = TF.adaptive_max_pool2d(o64, 1).flatten(1)
            torch.autograd.backward([o64] + ([f64] if want_head else []),
                                    [g_out.double() * (0.0 if last else 1.0)] + ([g_feat.double()] if want_head else []))
            return {k: q.grad for k, q in b64.named_parameters()}

        def perturbed_grads():
            """The fp32 CPU block again with the upstream gradient perturbed by 1e-6 relative (what two correct fp32 evaluations of
            the layers behind this block differ by): how far a parameter gradient moves under that is its conditioning."""
            import copy
            b32 = copy.deepcopy(blk)
            for q in b32.parameters():
                q.grad = None
            o32 = b32(x)
            f32 = TF.adaptive_max_pool2d(o32, 1).flatten(1)
            gp = torch.Generator().manual_seed(99)
            go = g_out * (1.0 + 1e-6 * torch.randn(g_out.shape, generator=gp))
            torch.autograd.backward([o32] + ([f32] if want_head else []), [go * (0.0 if last else 1.0)] + ([g_feat] if want_head else []))
            return {k: q.grad for k, q in b32.named_parameters()}
    g64 = None
    gpert = None
    worst = ("", 0.0)
    for name, got, want in pairs:
        got, want = got.detach().cpu().double(), want.detach().double()
        scale = max(1.0, float(want.abs().max())) if name.startswith("grad") or name.startswith("d") else 1.0
        d = (got - want).abs() / scale
        if float(d.max()) > worst[1]:
            worst = (name, float(d.max()))
        # inside one block a 2 x 2 pool window (or a global-max pick) whose two largest values differ by less than fp32
        # rounding may still go the other way (one flipped window moves the weight gradients of one output channel by
        # up to ~7e-3 of the tensor's scale: measured): on the large tensors allow up to 1 % of the elements beyond 1e-3 (measured <= 0.28 %),
        # none beyond 2e-2; every tensor within 1e-3 in rms (the worst are the first BatchNorm's dgamma / dbeta of blocks 0
        # and 1 -- fp32 sums over 220 k .. 5.5 M positions on both sides -- at 2 .. 5e-4 of their scale)
        assert float(d.max()) < 2e-2, (stage, name, float(d.max()))
        if d.numel() > 10000:
            assert float((d > 1e-3).double().mean()) < 1e-2, (stage, name, float((d > 1e-3).double().mean()))
        rms = float(d.pow(2).mean().sqrt())
        if rms >= 1e-3 and stage != "head" and name.startswith("grad "):
            # beyond 1e-3 of the fp32 CPU oracle: legitimate only where that oracle is itself that far from fp64
            if g64 is None:
                g64 = fp64_grads()
            ref64 = g64[name[5:]].detach()
            ours64 = float(((got - ref64) / scale).pow(2).mean().sqrt())
            cpu64 = float(((want - ref64) / scale).pow(2).mean().sqrt())
            _report("cfg2 stage %s %s: %.2e from the fp32 CPU oracle; against fp64: accelerated %.2e, fp32 CPU oracle %.2e | values: "
                    "accelerated %s, fp32 oracle %s, fp64 %s" % (stage, name, rms, ours64, cpu64, got.flatten()[:4].tolist(),
                                                                want.flatten()[:4].tolist(), ref64.flatten()[:4].tolist()))
            if gpert is None:
                gpert = perturbed_grads()
            moved = float(((gpert[name[5:]].detach().double() - want) / scale).pow(2).mean().sqrt())
            _report("cfg2 stage %s %s: the fp32 CPU oracle's own value moves by %.2e under a 1e-6 relative perturbation of the upstream gradient"
                    % (stage, name, moved))
            assert ours64 < max(1e-3, 2.0 * cpu64, 5.0 * moved), (stage, name, ours64, cpu64, moved)
            # ... and an ABSOLUTE cap whatever the oracle's own error is (a bound that only scales with the reference's
            # error lets a real regression of a few 1e-3 through): measured <= 1.1e-3 on every stage and arithmetic
            assert ours64 < 3e-3, (stage, name, ours64, cpu64, moved)
        else:
            assert rms < 1e-3, (stage, name, rms)
    _report("cfg2 stage %s on identical inputs: worst scaled difference %.2e (%s)" % (stage, worst[1], worst[0]))


# ------------------------------------------------------------------------------ split-fp16 where it can break
@pytest.fixture
def f16x3():
    mode0 = F.get_conv_arith()
    F.set_conv_arith(3)
    yield
    F.set_conv_arith(mode0)


@pytest.mark.parametrize("case", [(16, 100, 150, 16, 43, 3), (16, 150, 150, 8, 26, 1), (41, 64, 48, 1, 300, 3)])
def test_split_fp16_gradient_rows_spread_over_2_to_minus_40(case, f16x3):
    """`dout` whose per-sample rows are scaled by 2^0 .. 2^-40 (LSEP gives per-sample gradients of very different
    size).  One power-of-two scale serves the whole tensor, so the guarantee is ABSOLUTE: error <= a few fp32 ulps of
    the products of the LARGEST rows (what an fp32 sum over the batch loses anyway, and what the batch-statistics BN
    backward downstream mixes into every row); rows within 2^-10 of the maximum also keep fp32 RELATIVE accuracy.
    The per-row relative error is written to the report."""
    n, c_in, c_out, h, w, k = case
    kh = 1 if h == 1 else k
    pad = (kh // 2, k // 2)
    torch.manual_seed(sum(case))
    x = torch.randn(n, c_in, h, w)
    wt = torch.randn(c_out, c_in, kh, k) / (c_in * kh * k) ** 0.5
    expo = torch.linspace(0, -40, n).round()
    gy = torch.randn(n, c_out, h, w) * torch.exp2(expo).view(n, 1, 1, 1)
    dx64 = torch.nn.grad.conv2d_input(x.shape, wt.double(), gy.double(), padding=pad)
    dw64 = torch.nn.grad.conv2d_weight(x.double(), wt.shape, gy.double(), padding=pad)
    dx32 = torch.nn.grad.conv2d_input(x.shape, wt, gy, padding=pad)
    dw32 = torch.nn.grad.conv2d_weight(x, wt.shape, gy, padding=pad)
    d = F._desc(n, c_in, c_out, h, w, kh, k)
    assert F.plan_name(d, 1).startswith("conv_fwd_x3_kernel") and F.plan_name(d, 2).startswith("conv_wgrad"), (
        F.plan_name(d, 1), F.plan_name(d, 2))
    dx = F.conv_dgrad(gy.to(DEV), wt.to(DEV), x.shape).cpu()
    dw = F.conv_wgrad(x.to(DEV), gy.to(DEV), wt.shape).cpu()
    eps = 2.0 ** -23
    top = float(dx64.abs().max())
    e_abs = float((dx.double() - dx64).abs().max())
    assert e_abs < 8 * eps * top, (e_abs, top)
    rel = []
    for r in range(n):
        denom = float(dx64[r].abs().max())
        rel.append(float((dx[r].double() - dx64[r]).abs().max()) / denom)
        if expo[r] >= -10:
            assert rel[-1] < 16 * eps, (r, float(expo[r]), rel[-1])
    _report("row-scaled dgrad %s: abs err %.2e of max %.2e; per-row rel err at 2^[%s] = [%s]" % (
        case, e_abs, top, ", ".join("%d" % e for e in expo.tolist()), ", ".join("%.1e" % v for v in rel)))
    # the weight gradient sums the rows: dominated by the large ones, fp32-accurate in absolute terms
    e_dw = float((dw.double() - dw64).abs().max())
    e_dw32 = float((dw32.double() - dw64).abs().max())
    assert e_dw < 4.0 * e_dw32 + 8 * eps * float(dw64.abs().max()), (e_dw, e_dw32)
    assert float((dx32.double() - dx64).abs().max()) <= e_abs * 64 + 1.0      # (fp32 reference computed for the report)


@pytest.mark.parametrize("bad", [float("inf"), float("-inf"), float("nan")])
@pytest.mark.parametrize("case", [(2, 100, 100, 16, 43, 3), (2, 100, 100, 16, 43, 1)])
def test_split_fp16_non_finite_operands_surface(case, bad, f16x3):
    """An Inf or NaN in an operand must come out non-finite wherever F.conv2d's output is non-finite (never a finite
    number): NaN elements travel through the limbs; an Inf makes the declared maximum Inf, and the kernels then mark
    EVERY output of the call non-finite (include/fsc_hip.h)."""
    n, c_in, c_out, h, w, k = case
    torch.manual_seed(1)
    x = torch.randn(n, c_in, h, w)
    wt = torch.randn(c_out, c_in, k, k) / (c_in * k * k) ** 0.5
    gy = torch.randn(n, c_out, h, w)
    xb = x.clone()
    xb[1, 37, 5, 11] = bad
    ref = TF.conv2d(xb, wt, None, padding=k // 2)
    got = F.conv_forward(xb.to(DEV), wt.to(DEV), None).cpu()
    assert not torch.isfinite(got[~torch.isfinite(ref)]).any()
    if bad != bad:       # NaN: the rest of the output is untouched and still right
        ok = torch.isfinite(ref)
        assert float((got[ok] - ref[ok]).abs().max()) < 1e-4
    else:
        assert not torch.isfinite(got).any()
    gb = gy.clone()
    gb[0, 3, 2, 2] = bad
    rdx = torch.nn.grad.conv2d_input(x.shape, wt, gb, padding=k // 2)
    dx = F.conv_dgrad(gb.to(DEV), wt.to(DEV), x.shape).cpu()
    assert not torch.isfinite(dx[~torch.isfinite(rdx)]).any()
    rdw = torch.nn.grad.conv2d_weight(x, wt.shape, gb, padding=k // 2)
    dw = F.conv_wgrad(x.to(DEV), gb.to(DEV), wt.shape).cpu()
    assert not torch.isfinite(dw[~torch.isfinite(rdw)]).any()
    wb = wt.clone()
    wb[5, 7] = bad
    rw = TF.conv2d(x, wb, None, padding=k // 2)
    gw = F.conv_forward(x.to(DEV), wb.to(DEV), None).cpu()
    assert not torch.isfinite(gw[~torch.isfinite(rw)]).any()


def test_lsep_overflow_step_is_loudly_non_finite():
    """The reference's LSEP exponentiates score gaps unclamped (networks/losses.py:52): a gap above ~88 overflows.
    The accelerated step must surface that as a non-finite loss and non-finite gradients, exactly like the CPU path --
    never as finite garbage."""
    torch.manual_seed(2)
    exp = NS(config=NS(
        network=NS(num_conv_blocks=2, start_deep_supervision_on=0, conv_base_depth=64, growth_rate=1.5,
                   output_dropout=0.0, aggregation_type="max"),
        data=NS(features="mel_1024_512_64", _input_dim=64, _n_classes=80),
        train=NS(accumulation_steps=1, optimizer="adam", learning_rate=1e-3, weight_decay=0.0,
                 scheduler="1cycle_0.0001_0.005")))
    m = TwoDimensionalCNNClassificationModel(exp, device="cuda:0")
    ref = oref.TagCNN2d("mel_1024_512_64", 2, 64, 1.5, 0, 80)
    with torch.no_grad():
        m.output_transform[5].weight.mul_(400.0)           # logits of several hundred -> exp() overflows in fp32
    ref.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    signal = 0.1 * torch.randn(8, 20000, 1)
    labels = torch.zeros(8, 80)
    labels[torch.arange(8), torch.randint(0, 80, (8,))] = 1.0
    ref.train()
    rl = ref(signal)["class_logits"]
    rper = oref.lsep(rl, labels, average=False)
    rper.mean().backward()
    assert not torch.isfinite(rper).all()                  # the CPU path overflows on this input
    m.train()
    m.make_optimizer(max_steps=10)
    logits, per, loss = m.training_step(signal.to(DEV), labels.to(DEV), step_optimizer=False)
    assert float((logits.cpu() - rl).abs().max()) < 1e-2 * float(rl.abs().max())
    assert torch.equal(torch.isfinite(per.cpu()), torch.isfinite(rper))
    assert not torch.isfinite(loss)
    rg = dict(ref.named_parameters())
    for k, p in m.named_parameters():
        if not torch.isfinite(rg[k].grad).all():
            assert not torch.isfinite(p.grad).all(), k
