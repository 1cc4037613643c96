"""Generate golden vectors by importing the REFERENCE itself (read-only) from /root/reference.

Runs only in the build container (the reference never travels to the GPU box).  Output:
small .npz / .json fixtures next to this file -- inputs and expected outputs, no reference
source.  Harness recipe = SURVEY.md Appendix A: stub the absent third-party modules, shim
torch.stft's removed real-pair output, then import the reference modules unmodified.

    python tests/golden/make_golden.py
"""
import json
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

from oracle import mel as oracle_mel  # noqa: E402  (librosa restatement; librosa is absent)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _NullWriter:
    def __init__(self, *a, **k):
        pass

    def add_scalar(self, *a, **k):
        pass

    add_image = add_histogram = add_scalar


def _librosa_mel(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    return oracle_mel.slaney_mel_filterbank(sr, n_fft, n_mels, fmin=fmin, fmax=fmax)


_stub("umap")
_stub("pysndfx")
_stub("iterstrat")
_stub("iterstrat.ml_stratifiers", MultilabelStratifiedKFold=object)
_stub("torchvision").utils = _stub("torchvision.utils", make_grid=lambda *a, **k: None)
_stub("tensorboardX", SummaryWriter=_NullWriter)
_stub("pretrainedmodels")
_stub("pretrainedmodels.models", resnet18=None, resnet34=None)
_lib = _stub("librosa")
_lib.effects = _stub("librosa.effects")
_lib.filters = _stub("librosa.filters", mel=_librosa_mel)

_orig_stft = torch.stft


def _stft_real_pair(x, n_fft, **k):
    k["return_complex"] = True
    return torch.view_as_real(_orig_stft(x, n_fft, **k))


torch.stft = _stft_real_pair

from networks.classifiers import (  # noqa: E402
    TwoDimensionalCNNClassificationModel, HierarchicalCNNClassificationModel)
from networks.losses import lsep_loss, binary_cross_entropy  # noqa: E402
from ops.utils import lwlrap, compute_torch_stft, make_mel_filterbanks  # noqa: E402
from ops.audio import mix_audio_and_labels  # noqa: E402
from ops.padding import make_collate_fn, BucketingSampler  # noqa: E402
from ops.training import OPTIMIZERS, make_scheduler, make_step  # noqa: E402


class NS(dict):
    __getattr__ = dict.__getitem__


def experiment(features, blocks, base, growth, start, input_dim, dropout=0.0, n_classes=80,
               optimizer="adam", lr=1e-3, wd=0.0, scheduler="1cycle_0.0001_0.005"):
    return NS(config=NS(
        network=NS(num_conv_blocks=blocks, start_deep_supervision_on=start,
                   conv_base_depth=base, growth_rate=growth, output_dropout=dropout,
                   aggregation_type="max"),
        data=NS(features=features, _input_dim=input_dim, _n_classes=n_classes),
        train=NS(accumulation_steps=1, optimizer=optimizer, learning_rate=lr,
                 weight_decay=wd, scheduler=scheduler)))


def seed_all(s):
    torch.manual_seed(s)
    np.random.seed(s)
    random.seed(s)


def np_state(model):
    return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def labels_multi_hot(n, c, rng, extra=0.02):
    y = (rng.random((n, c)) < extra).astype(np.float32)
    y[np.arange(n), rng.integers(0, c, n)] = 1.0
    return y


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024))


# ---------------------------------------------------------------------------------- G1/G2
def g1_frontend():
    out = {}
    seed_all(1)
    cases = {
        "mel_1024_512_64": (3, 16000),
        "mel_2048_1024_128": (2, 22050),
        "stft_256_128": (2, 8000),
    }
    for desc, (n, t) in cases.items():
        wav = (0.1 * torch.randn(n, t)).float()
        wav[-1, t // 2:] = 0.0                       # a zero-padded row (collate tail)
        mag = compute_torch_stft(wav, desc)
        out[desc + ".wav"] = wav.numpy()
        out[desc + ".mag"] = mag.numpy()
        if desc.startswith("mel"):
            fb = torch.from_numpy(make_mel_filterbanks(desc))
            logmel = torch.log(torch.nn.functional.conv1d(mag, fb.unsqueeze(-1)) + 1e-4)
            out[desc + ".logmel"] = logmel.numpy()
        else:
            out[desc + ".logmag"] = torch.log(mag + 1e-4).numpy()
    save("g1_frontend.npz", **out)
    fbs = {d: make_mel_filterbanks(d) for d in ("mel_1024_512_64", "mel_2048_1024_128")}
    save("g2_filterbanks.npz", **fbs)


# ---------------------------------------------------------------------------------- G15
def lcg_noise(n, seed):
    """Deterministic noise in [-0.5, 0.5) from a 31-bit linear congruential generator in integer arithmetic: the tests rebuild the
    waveform from (n, seed) instead of storing 1.7 MB of it (tests/test_oracle_cpu.py lcg_noise is the same function)."""
    out = np.empty(n, dtype=np.float64)
    state = np.uint64(seed)
    a, c, mask = np.uint64(1103515245), np.uint64(12345), np.uint64((1 << 31) - 1)
    for i in range(n):
        state = (a * state + c) & mask
        out[i] = float(state) / float(1 << 31) - 0.5
    return out.astype(np.float32)


G15_FRAMES = np.r_[0:4, 1000:1004, 1720:1728, 2340:2352, 3440:3446]


def g15_frontend_10s():
    """compute_torch_stft of the reference (ops/utils.py:110-127) with stft_256_128 on 10 s @ 44.1 kHz clips -- the cfg-3 front-end
    shape: 441 000 samples -> 129 x 3446 frames.  Clip 1 has a zero tail from sample 300 000.  Stored: log(|stft| + 1e-4) at 34
    frames and fp64 checksums of all frames."""
    t = 441000
    wav = np.stack([0.2 * lcg_noise(t, 15), 0.2 * lcg_noise(t, 16)])
    wav[1, 300000:] = 0.0
    mag = compute_torch_stft(torch.from_numpy(wav), "stft_256_128")
    logmag = torch.log(mag + 1e-4)
    assert tuple(logmag.shape) == (2, 129, 3446)
    save("g15_frontend_10s.npz", seeds=np.array([15, 16]), t=np.int64(t), scale=np.float32(0.2), zero_from=np.int64(300000),
         frames=G15_FRAMES.astype(np.int64), logmag=logmag[:, :, G15_FRAMES].numpy(),
         sum=logmag.double().sum((1, 2)).numpy(), abs_sum=logmag.double().abs().sum((1, 2)).numpy(),
         mag_sum=mag.double().sum((1, 2)).numpy())


# ---------------------------------------------------------------------------------- G3/G9
def _model_case(model, signal, labels, n_adam_steps, prefix, out, average=False):
    """state_dict, logits, per-sample LSEP, all param grads, BN stats after step 1, and the
    parameters after `n_adam_steps` Adam-amsgrad steps; then eval-mode logits."""
    for k, v in np_state(model).items():
        out[prefix + "init." + k] = v
    out[prefix + "signal"] = signal.numpy()
    out[prefix + "labels"] = labels.numpy()
    model.train()
    model.make_optimizer(max_steps=100)
    for step in range(n_adam_steps):
        make_step(model.scheduler, step=step + 1)
        out[prefix + "lr.%d" % step] = np.float64(model.optimizer.param_groups[0]["lr"])
        model.optimizer.zero_grad()
        logits = model(signal)["class_logits"]
        if average:
            loss = lsep_loss(logits.squeeze(), labels)
            per = loss.detach().reshape(1)
            loss.backward()
        else:
            per = lsep_loss(logits, labels, average=False)
            per.mean().backward()
        if step == 0:
            out[prefix + "logits"] = logits.detach().numpy()
            out[prefix + "loss"] = per.detach().numpy()
            for k, p in model.named_parameters():
                out[prefix + "grad." + k] = p.grad.detach().numpy().copy()
            # eval-mode forward with the INITIAL parameters and the BN running statistics
            # after exactly one update (taken before optimizer.step: parameters whose true
            # gradient is zero get rounding-noise Adam updates)
            model.eval()
            with torch.no_grad():
                ev = model(signal)["class_logits"]
                out[prefix + "eval_logits"] = ev.numpy()
                out[prefix + "eval_probs"] = torch.sigmoid(ev).numpy()
            model.train()
            for k, v in np_state(model).items():
                if "running" in k or "num_batches" in k:
                    out[prefix + "bn1." + k] = v
        model.optimizer.step()
    for k, v in np_state(model).items():
        out[prefix + "final." + k] = v


def g3_tiny2d():
    seed_all(3)
    exp = experiment("mel_1024_512_64", blocks=2, base=8, growth=1.5, start=1, input_dim=64,
                     lr=1e-3)
    model = TwoDimensionalCNNClassificationModel(exp, device="cpu")
    rng = np.random.default_rng(3)
    signal = (0.1 * torch.randn(4, 16000, 1)).float()
    signal[-1, 9000:] = 0.0
    labels = torch.from_numpy(labels_multi_hot(4, 80, rng))
    out = {}
    _model_case(model, signal, labels, 3, "", out)
    save("g3_tiny2d.npz", **out)
    with open(os.path.join(HERE, "g3_state_keys.json"), "w") as f:
        json.dump([[k, list(v.shape), str(v.dtype)] for k, v in model.state_dict().items()], f,
                  indent=0)


def g3b_three_block():
    """3 blocks, start 0 (head on every block), odd channel counts, dropout 0."""
    seed_all(33)
    exp = experiment("mel_1024_512_64", blocks=3, base=10, growth=1.5, start=0, input_dim=64)
    model = TwoDimensionalCNNClassificationModel(exp, device="cpu")
    rng = np.random.default_rng(33)
    signal = (0.1 * torch.randn(3, 20000, 1)).float()
    labels = torch.from_numpy(labels_multi_hot(3, 80, rng))
    out = {}
    _model_case(model, signal, labels, 1, "", out)
    save("g3b_threeblock2d.npz", **out)


# ---------------------------------------------------------------------------------- G4
def g4_block():
    seed_all(4)
    exp = experiment("mel_1024_512_64", blocks=2, base=12, growth=1.5, start=0, input_dim=64)
    model = TwoDimensionalCNNClassificationModel(exp, device="cpu")
    out = {}
    for idx, (c_in, h, w) in enumerate([(2, 13, 21), (12, 9, 7)]):
        blk = model.conv_modules[idx]
        blk.train()
        x = torch.randn(3, c_in, h, w, requires_grad=True)
        y = blk(x)
        g = torch.randn_like(y)
        y.backward(g)
        p = "blk%d." % idx
        for k, v in blk.state_dict().items():
            out[p + "state." + k] = v.detach().numpy().copy()   # after the fwd (BN stats moved)
        out[p + "x"] = x.detach().numpy()
        out[p + "y"] = y.detach().numpy()
        out[p + "gy"] = g.numpy()
        out[p + "gx"] = x.grad.numpy()
        for k, prm in blk.named_parameters():
            out[p + "grad." + k] = prm.grad.numpy().copy()
    # initial parameters (BN buffers at their init values are implied: mean 0, var 1)
    save("g4_block.npz", **out)


# ---------------------------------------------------------------------------------- G5
def g5_tiny1d():
    seed_all(5)
    exp = experiment("stft_256_128", blocks=3, base=12, growth=1.5, start=1, input_dim=129)
    model = HierarchicalCNNClassificationModel(exp, device="cpu")
    rng = np.random.default_rng(5)
    signal = (0.1 * torch.randn(4, 8000, 1)).float()
    signal[-1, 5000:] = 0.0
    labels = torch.from_numpy(labels_multi_hot(4, 80, rng))
    out = {}
    _model_case(model, signal, labels, 2, "", out, average=True)
    save("g5_tiny1d.npz", **out)
    with open(os.path.join(HERE, "g5_state_keys.json"), "w") as f:
        json.dump([[k, list(v.shape), str(v.dtype)] for k, v in model.state_dict().items()], f,
                  indent=0)


# ---------------------------------------------------------------------------------- G6
def g6_losses():
    seed_all(6)
    rng = np.random.default_rng(6)
    out = {}
    logits = (2.0 * torch.randn(16, 80)).requires_grad_()
    y = np.zeros((16, 80), np.float32)
    for r in range(16):
        y[r, rng.choice(80, size=1 + r % 4, replace=False)] = 1.0
    y = torch.from_numpy(y)
    for avg in (True, False):
        logits.grad = None
        val = lsep_loss(logits, y, average=avg)
        (val if avg else val.sum()).backward()
        tag = "avg" if avg else "per"
        out["lsep_%s" % tag] = val.detach().numpy()
        out["lsep_%s_grad" % tag] = logits.grad.numpy().copy()
    # general (non-binary) targets pin the pairwise mask t_j < t_i
    soft = torch.from_numpy(rng.choice([0.0, 0.5, 1.0], size=(16, 80), p=[0.9, 0.05, 0.05])
                            .astype(np.float32))
    logits.grad = None
    val = lsep_loss(logits, soft, average=False)
    val.sum().backward()
    out["soft_targets"] = soft.numpy()
    out["lsep_soft"] = val.detach().numpy()
    out["lsep_soft_grad"] = logits.grad.numpy().copy()
    logits.grad = None
    val = binary_cross_entropy(logits, y)
    val.backward()
    out["bce"] = val.detach().numpy()
    out["bce_grad"] = logits.grad.numpy().copy()
    out["logits"] = logits.detach().numpy()
    out["targets"] = y.numpy()
    save("g6_losses.npz", **out)


# ---------------------------------------------------------------------------------- G7
def g7_mixup():
    out = {}
    rng = np.random.default_rng(7)
    cases = {"long_first": (5000, 3000), "short_first": (2500, 4100), "equal": (3000, 3000)}
    for name, (la, lb) in cases.items():
        a = rng.standard_normal(la).astype(np.float32)
        b = rng.standard_normal(lb).astype(np.float32)
        ya = labels_multi_hot(1, 80, rng)[0]
        yb = labels_multi_hot(1, 80, rng)[0]
        out[name + ".a"], out[name + ".b"] = a.copy(), b.copy()
        out[name + ".ya"], out[name + ".yb"] = ya.copy(), yb.copy()
        np.random.seed(70)
        random.seed(70)
        mixed, y = mix_audio_and_labels(a, b, ya, yb)
        out[name + ".mixed"] = np.asarray(mixed)
        out[name + ".labels"] = y
    save("g7_mixup.npz", **out)


# ---------------------------------------------------------------------------------- G8
def g8_collate_bucketing():
    rng = np.random.default_rng(8)
    out = {}
    lens = [700, 1000, 320, 999]
    batch = [dict(signal=rng.standard_normal((n, 1)).astype(np.float32),
                  labels=labels_multi_hot(1, 80, rng)[0],
                  is_noisy=np.float64(i % 2)) for i, n in enumerate(lens)]
    for i, s in enumerate(batch):
        out["collate.in%d" % i] = s["signal"].copy()
    col = make_collate_fn({"signal": 0.0})(batch)
    out["collate.signal"] = col["signal"].numpy()
    out["collate.labels"] = col["labels"].numpy()
    out["collate.is_noisy"] = col["is_noisy"].numpy()
    out["collate.dtypes"] = np.array([str(col[k].dtype) for k in ("signal", "labels", "is_noisy")])

    class DS:
        pass

    results = {}
    ds = DS()
    ds.lengths = np.array(list(range(1, 13)) + [50, 100])
    random.seed(0)
    s = BucketingSampler(ds, 10, [0, 4, 8, 16, 64])
    results["small"] = dict(lengths=[int(x) for x in ds.lengths], max_batch_elems=10,
                            buckets=[0, 4, 8, 16, 64], seed=0,
                            batches=[[int(i) for i in b] for b in s])
    ds = DS()
    ds.lengths = rng.integers(13230, 1323000, size=500)
    random.seed(11)
    edges = [0] + [88200 * k for k in range(1, 16)]
    s = BucketingSampler(ds, 128 * 441000 // 16, edges)
    results["large"] = dict(lengths=[int(x) for x in ds.lengths],
                            max_batch_elems=128 * 441000 // 16, buckets=edges, seed=11,
                            batches=[[int(i) for i in b] for b in s])
    with open(os.path.join(HERE, "g8_bucketing.json"), "w") as f:
        json.dump(results, f)
    save("g8_collate.npz", **out)


# ---------------------------------------------------------------------------------- G9
def g9_schedules():
    out = {}
    p = [torch.nn.Parameter(torch.zeros(1))]
    for name, (sched, steps) in {"a": ("1cycle_0.0001_0.005", 100),
                                 "b": ("1cycle_0.001_0.01", 37)}.items():
        opt = OPTIMIZERS["adam"](p, 1e-3)
        sch = make_scheduler(sched, max_steps=steps)(opt)
        trace = []
        for s in range(steps):
            make_step(sch, step=s + 1)
            trace.append(opt.param_groups[0]["lr"])
        out["onecycle_" + name] = np.array(trace, np.float64)
    # SGD-nesterov + weight decay trace on a small vector
    seed_all(9)
    w = torch.nn.Parameter(torch.randn(257))
    out["sgd.w0"] = w.detach().numpy().copy()
    opt = OPTIMIZERS["momentum"]([w], 0.05, weight_decay=1e-3)
    for s in range(3):
        g = torch.randn(257)
        out["sgd.g%d" % s] = g.numpy().copy()
        w.grad = g.clone()
        opt.step()
        out["sgd.w%d" % (s + 1)] = w.detach().numpy().copy()
    w = torch.nn.Parameter(torch.randn(257))
    out["adam.w0"] = w.detach().numpy().copy()
    opt = OPTIMIZERS["adam"]([w], 0.003, weight_decay=1e-2)
    for s in range(4):
        g = torch.randn(257) * (1.0 if s != 2 else 0.01)      # step 2 keeps vmax > v (amsgrad)
        out["adam.g%d" % s] = g.numpy().copy()
        w.grad = g.clone()
        opt.step()
        out["adam.w%d" % (s + 1)] = w.detach().numpy().copy()
    save("g9_optim.npz", **out)


# ---------------------------------------------------------------------------------- G10
def g10_lwlrap():
    rng = np.random.default_rng(10)
    truth = labels_multi_hot(32, 80, rng, extra=0.03)
    truth[5] = 0.0                                             # a row with no positives
    truth[9] = 1.0                                             # a row with all positives
    scores = rng.random((32, 80)).astype(np.float32)
    scores[:, ::7] = np.round(scores[:, ::7], 1)               # ties
    scores[3] = 0.5                                            # one fully tied row
    val = lwlrap(truth, scores)
    save("g10_lwlrap.npz", truth=truth, scores=scores, value=np.float64(val))


# ---------------------------------------------------------------------------------- G11
def g11_cfg1():
    """cfg 1: 64 synthetic 2 s @ 16 kHz clips, mel_1024_512_64, 3 blocks base 32 growth 2,
    2 training steps (Adam-amsgrad, 1cycle); logits + loss trace for both LSEP and BCE."""
    out = {}
    rng = np.random.default_rng(11)
    gen = torch.Generator().manual_seed(1234)
    signal = 0.1 * torch.randn(64, 32000, 1, generator=gen)
    labels = torch.from_numpy(labels_multi_hot(64, 80, rng))
    out["signal_seed"] = np.int64(1234)
    out["labels"] = labels.numpy()
    for loss_name in ("lsep", "bce"):
        seed_all(11)
        exp = experiment("mel_1024_512_64", blocks=3, base=32, growth=2, start=1, input_dim=64,
                         lr=1e-3)
        model = TwoDimensionalCNNClassificationModel(exp, device="cpu")
        model.train()
        model.make_optimizer(max_steps=10)
        for step in range(2):
            make_step(model.scheduler, step=step + 1)
            model.optimizer.zero_grad()
            logits = model(signal)["class_logits"]
            if loss_name == "lsep":
                per = lsep_loss(logits, labels, average=False)
                loss = per.mean()
            else:
                loss = binary_cross_entropy(logits, labels)
            loss.backward()
            model.optimizer.step()
            out["%s.logits%d" % (loss_name, step)] = logits.detach().numpy()
            out["%s.loss%d" % (loss_name, step)] = np.float64(loss.item())
            probs = torch.sigmoid(logits).detach().numpy()
            out["%s.lwlrap%d" % (loss_name, step)] = np.float64(lwlrap(labels.numpy(), probs))
    out["n_params"] = np.int64(sum(p.numel() for p in model.parameters()))
    save("g11_cfg1.npz", **out)


# ---------------------------------------------------------------------------------- G12
def g12_cfg2_step():
    """cfg 2 of BASELINE.json at its real widths (6 blocks, base 100, growth 1.5 -> 100/150/225/337/506/759
    channels, mel_2048_1024_128, 10 s @ 44.1 kHz), batch 4, one training forward/backward of the REFERENCE
    model.  The 86 MB state dict is not stored: the product registers the same modules in the same order, so
    `torch.manual_seed(seed)` reproduces the reference's initial parameters bit for bit (checked through the
    per-parameter checksums stored here).  Stored: logits, per-sample LSEP, eval-mode logits after the one
    BN update, and per parameter the gradient's l2 norm, sum and 256 elements at seeded positions."""
    seed = 2024
    seed_all(seed)
    exp = experiment("mel_2048_1024_128", blocks=6, base=100, growth=1.5, start=1, input_dim=128)
    model = TwoDimensionalCNNClassificationModel(exp, device="cpu")
    gen = torch.Generator().manual_seed(4321)
    signal = 0.1 * torch.randn(4, 441000, 1, generator=gen)
    signal[-1, 300000:] = 0.0                                  # collate tail
    rng = np.random.default_rng(12)
    labels = torch.from_numpy(labels_multi_hot(4, 80, rng))
    out = {"seed": np.int64(seed), "signal_seed": np.int64(4321), "labels": labels.numpy(),
           "zero_tail_from": np.int64(300000)}
    names = []
    for k, p in model.named_parameters():
        names.append(k)
        v = p.detach().double()
        out["init_sum." + k] = np.float64(v.sum().item())
        out["init_abs." + k] = np.float64(v.abs().sum().item())
    out["n_params"] = np.int64(sum(p.numel() for p in model.parameters()))
    model.train()
    logits = model(signal)["class_logits"]
    per = lsep_loss(logits, labels, average=False)
    per.mean().backward()
    out["logits"] = logits.detach().numpy()
    out["loss"] = per.detach().numpy()
    pick = np.random.default_rng(99)
    for k, p in model.named_parameters():
        g = p.grad.detach().reshape(-1)
        idx = pick.integers(0, g.numel(), size=min(256, g.numel()))
        out["grad_idx." + k] = idx.astype(np.int64)
        out["grad_val." + k] = g[torch.from_numpy(idx)].numpy().copy()
        out["grad_norm." + k] = np.float64(g.double().norm().item())
        out["grad_sum." + k] = np.float64(g.double().sum().item())
        out["grad_absmax." + k] = np.float64(g.abs().max().item())
    model.eval()
    with torch.no_grad():
        out["eval_logits"] = model(signal)["class_logits"].numpy()
    save("g12_cfg2_step.npz", **out)


# ---------------------------------------------------------------------------------- G13
def g13_input_pipeline():
    """The reference's per-sample training transforms that the build moves to the device:
    Compose([loader, SampleLongAudio, MapLabels, ShuffleAudio, MixUp]) through the reference's own SoundDataset
    (random_clean_sample -> clean_transform), on seeded-noise clips (LoadAudio needs librosa + files, so a loader with
    the same protocol returns 0.1 * N(0, 1) drawn from the clip's seed).  Stored: clip descriptors, generator seeds,
    and per sample the resulting waveform and labels, in dataset order."""
    from datasets.sound_dataset import SoundDataset
    from ops.transforms import Compose, MapLabels, MixUp, SampleLongAudio, ShuffleAudio

    class SeededNoise:
        def __call__(self, dataset, **inputs):
            out = dict(inputs)
            _, seed, n, sr = str(inputs["filename"]).split(":")
            out["audio"] = (0.1 * np.random.RandomState(int(seed)).standard_normal(int(n))).astype(np.float32)
            out["sr"] = int(sr)
            return out

    sr = 8000
    rng = np.random.default_rng(13)
    lengths = rng.integers(int(0.3 * sr), int(3.2 * sr), size=14)
    lengths[3] = lengths[7]                                   # an equal-length MixUp pair is possible
    files = ["synthetic:%d:%d:%d" % (100 + i, n, sr) for i, n in enumerate(lengths)]
    class_map = {"c%02d" % i: i for i in range(80)}
    labels = [["c%02d" % rng.integers(80)] + (["c%02d" % rng.integers(80)] if i % 3 == 0 else []) for i in range(14)]
    out = {"files": np.array(files), "labels_csv": np.array([",".join(l) for l in labels]), "sr": np.int64(sr),
           "max_length": np.int64(1), "chunk_length": np.float64(0.25), "p_shuffle": np.float64(0.6),
           "p_mixup": np.float64(0.5), "seed": np.int64(131)}
    clean = Compose([SeededNoise(), SampleLongAudio(max_length=1), MapLabels(class_map=class_map)])
    tf = Compose([SeededNoise(), SampleLongAudio(max_length=1), MapLabels(class_map=class_map),
                  ShuffleAudio(chunk_length=0.25, p=0.6), MixUp(p=0.5)])
    ds = SoundDataset(audio_files=files, labels=labels, transform=tf, clean_transform=clean)
    seed_all(131)
    for i in range(len(ds)):
        s = ds[i]
        out["audio.%d" % i] = np.asarray(s["audio"], np.float32).copy()
        out["labels.%d" % i] = np.asarray(s["labels"], np.float32).copy()
    save("g13_input_pipeline.npz", **out)


# ---------------------------------------------------------------------------------- G14
def g14_rnn_head():
    """aggregation_type="rnn" (LayerNorm + bi-GRU(128) on the frequency-averaged block output, classifiers.py:514-522,
    592-597): tiny 2-d model, 2 blocks, one RNN head; state-dict keys, initial parameters, logits, per-sample LSEP and
    every parameter gradient of the reference."""
    seed_all(14)
    exp = experiment("mel_1024_512_64", blocks=2, base=8, growth=1.5, start=1, input_dim=64)
    exp.config.network["aggregation_type"] = "rnn"
    model = TwoDimensionalCNNClassificationModel(exp, device="cpu")
    rng = np.random.default_rng(14)
    signal = (0.1 * torch.randn(3, 12000, 1)).float()
    signal[-1, 7000:] = 0.0
    labels = torch.from_numpy(labels_multi_hot(3, 80, rng))
    out = {"signal": signal.numpy(), "labels": labels.numpy(), "seed": np.int64(14)}
    for k, v in np_state(model).items():        # same seed + same registration order reproduce the parameters: checksums only
        if v.dtype == np.float32:
            out["init_sum." + k] = np.float64(v.astype(np.float64).sum())
            out["init_abs." + k] = np.float64(np.abs(v.astype(np.float64)).sum())
    model.train()
    logits = model(signal)["class_logits"]
    per = lsep_loss(logits, labels, average=False)
    per.mean().backward()
    out["logits"] = logits.detach().numpy()
    out["loss"] = per.detach().numpy()
    pick = np.random.default_rng(140)
    for k, p in model.named_parameters():
        g = p.grad.detach().reshape(-1)
        if g.numel() <= 4096:
            out["grad." + k] = g.numpy().copy()
        else:                                    # the GRU matrices: 4096 seeded elements + norm
            idx = pick.integers(0, g.numel(), size=4096)
            out["grad_idx." + k] = idx.astype(np.int64)
            out["grad." + k] = g[torch.from_numpy(idx)].numpy().copy()
        out["grad_norm." + k] = np.float64(g.double().norm().item())
    model.eval()
    with torch.no_grad():
        out["eval_logits"] = model(signal)["class_logits"].numpy()
    save("g14_rnn_head.npz", **out)
    with open(os.path.join(HERE, "g14_state_keys.json"), "w") as f:
        json.dump([[k, list(v.shape), str(v.dtype)] for k, v in model.state_dict().items()], f, indent=0)


if __name__ == "__main__":
    torch.set_num_threads(8)
    g1_frontend()
    g3_tiny2d()
    g3b_three_block()
    g4_block()
    g5_tiny1d()
    g6_losses()
    g7_mixup()
    g8_collate_bucketing()
    g9_schedules()
    g10_lwlrap()
    g11_cfg1()
    g12_cfg2_step()
    g13_input_pipeline()
    g14_rnn_head()
    g15_frontend_10s()
