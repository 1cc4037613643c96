"""ORACLE (test infrastructure, not product code).

Pure-PyTorch CPU restatement of the reference's audio-tagging hot path, written from the
behaviour documented in SURVEY.md section 8(a).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product package never does (it fails loudly without its HIP library).

Pinned: tests/test_oracle_cpu.py checks every function here against golden vectors that
tests/golden/make_golden.py produced by importing the reference itself from
/root/reference in the build container (the reference has no tests of its own).

Reference locations restated (all relative to /root/reference):
  stft_magnitude        ops/utils.py:110-127   (torch-1.0.1 stft defaults: center, reflect,
                                                 onesided, periodic hann, real-pair output)
  log_mel / log_stft    networks/classifiers.py:565-579 (2-d), :178-191 (1-d)
  add_frequency_channel networks/classifiers.py:553-561
  residual_unit         networks/classifiers.py:37-69 (1-d), :72-104 (2-d)
  TagCNN2d              networks/classifiers.py:485-551, 563-607
  TagCNN1d              networks/classifiers.py:109-217
  lsep / bce            networks/losses.py:47-58, :19-22
  train_step            networks/classifiers.py:652-690
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import mel as _mel

LOG_EPS = 1e-4


def parse_descriptor(descriptor):
    parts = descriptor.split("_")
    kind = parts[0]
    nums = [int(p) for p in parts[1:]]
    return kind, nums


def stft_magnitude(audio, descriptor):
    """(N, T) f32 -> (N, n_fft/2+1, 1 + T//hop) magnitude.  ops/utils.py:110-127."""
    _, nums = parse_descriptor(descriptor)
    n_fft, hop = nums[0], nums[1]
    win = torch.hann_window(n_fft, dtype=audio.dtype, device=audio.device)
    spec = torch.stft(audio, n_fft, hop_length=hop, window=win, center=True,
                      pad_mode="reflect", normalized=False, onesided=True,
                      return_complex=True)
    pair = torch.view_as_real(spec)
    return pair.pow(2).sum(-1).sqrt()


def features_from_signal(signal, descriptor, filterbank=None):
    """(N, T, 1) waveform -> (N, F, frames) log features (classifiers.py:565-579)."""
    kind, _ = parse_descriptor(descriptor)
    x = signal.squeeze(-1)
    if kind in ("stft", "mel"):
        x = stft_magnitude(x, descriptor)
        if kind == "stft":
            x = torch.log(x + LOG_EPS)
    if kind == "mel":
        x = torch.einsum("mf,nft->nmt", filterbank, x)
        x = torch.log(x + LOG_EPS)
    return x


def add_frequency_channel(x):
    """(N,1,H,W) -> (N,2,H,W) with a -1..1 ramp along H (classifiers.py:553-561)."""
    n, _, h, w = x.shape
    ramp = torch.linspace(-1, 1, h, device=x.device, dtype=x.dtype).view(1, 1, h, 1)
    return torch.cat([x, ramp.expand(n, 1, h, w)], dim=1)


def _residual_unit(depth, dims):
    conv = nn.Conv2d if dims == 2 else nn.Conv1d
    bn = nn.BatchNorm2d if dims == 2 else nn.BatchNorm1d

    class Unit(nn.Module):
        def __init__(self):
            super().__init__()
            # registration order matters: it fixes both the state-dict key order and the
            # RNG draw order of the default initialisers.
            self.conv1 = conv(depth, depth, 1)
            self.bn1 = bn(depth)
            self.conv2 = conv(depth, depth, 3, padding=1)
            self.bn2 = bn(depth)
            self.conv3 = conv(depth, depth, 1)
            self.bn3 = bn(depth)
            self.prelu1 = nn.PReLU(depth)
            self.prelu2 = nn.PReLU(depth)
            self.prelu3 = nn.PReLU(depth)

        def forward(self, x):
            y = self.prelu1(self.bn1(self.conv1(x)))
            y = self.prelu2(self.bn2(self.conv2(y)))
            y = self.bn3(self.conv3(y))
            return self.prelu3(y + x)

    return Unit()


def _block(c_in, depth, dims):
    if dims == 2:
        layers = [nn.BatchNorm2d(c_in), nn.Conv2d(c_in, depth, 3, padding=1),
                  nn.MaxPool2d(2, 2), nn.BatchNorm2d(depth), nn.PReLU(depth)]
    else:
        layers = [nn.BatchNorm1d(c_in), nn.Conv1d(c_in, depth, 3, padding=1),
                  nn.MaxPool1d(2, 2), nn.BatchNorm1d(depth), nn.PReLU(depth)]
    layers.append(_residual_unit(depth, dims))
    return nn.Sequential(*layers)


def block_depths(num_blocks, base, growth):
    return [int(growth ** k * base) for k in range(num_blocks)]


class _TagCNN(nn.Module):
    dims = 2

    def __init__(self, features, num_conv_blocks, conv_base_depth, growth_rate,
                 start_deep_supervision_on, n_classes, output_dropout=0.0, input_dim=None,
                 filterbank=None, aggregation_type="max"):
        super().__init__()
        self.features = features
        self.start = start_deep_supervision_on
        self.aggregation = aggregation_type
        self.filterbank = None
        if features.startswith("mel"):
            fb = filterbank if filterbank is not None else _mel.make_mel_filterbanks(features)
            self.filterbank = torch.as_tensor(fb)
        depths = block_depths(num_conv_blocks, conv_base_depth, growth_rate)
        first_in = 2 if self.dims == 2 else input_dim
        self.conv_modules = nn.ModuleList()
        self.rnns = nn.ModuleList()
        c_in = first_in
        total = 0
        for k, d in enumerate(depths):
            if k >= self.start:
                if aggregation_type == "max":
                    total += d
                else:     # classifiers.py:514-522 (2-d), :137-145 (1-d): created before the block's conv modules
                    total += 256
                    self.rnns.append(nn.Sequential(nn.LayerNorm((d,)),
                                                   nn.GRU(d, 128, batch_first=True, bidirectional=True)))
            self.conv_modules.append(_block(c_in, d, self.dims))
            c_in = d
        self.output_transform = nn.Sequential(
            nn.BatchNorm1d(total), nn.Linear(total, total), nn.BatchNorm1d(total),
            nn.PReLU(total), nn.Dropout(p=output_dropout), nn.Linear(total, n_classes))

    def front_end(self, signal):
        x = features_from_signal(signal, self.features, self.filterbank)
        if self.dims == 2:
            x = add_frequency_channel(x.unsqueeze(1))
        return x

    def trunk(self, x, collect=None):
        pooled = []
        for k, blk in enumerate(self.conv_modules):
            x = blk(x)
            if collect is not None:
                collect.append(x)
            if k >= self.start and self.aggregation == "rnn":
                # classifiers.py:592-597 (2-d: mean over frequency first), :202-207 (1-d)
                seq = (torch.mean(x, 2) if self.dims == 2 else x).permute(0, 2, 1)
                _, state = self.rnns[k - self.start](seq)
                pooled.append(state.permute(1, 0, 2).contiguous().view(seq.size(0), -1))
            elif k >= self.start:
                # AdaptiveMaxPool2d(1) / 1d(1) (classifiers.py:540,591 / :163,201): ONE arg-max per plane gets the
                # gradient (amax would split it between ties)
                pooled.append(F.adaptive_max_pool1d(x.flatten(2), 1).squeeze(-1))
        return torch.cat(pooled, dim=-1)

    def forward(self, signal):
        feats = self.trunk(self.front_end(signal))
        return {"class_logits": self.output_transform(feats)}


class TagCNN2d(_TagCNN):
    dims = 2


class TagCNN1d(_TagCNN):
    dims = 1


def lsep(logits, targets, average=True):
    """losses.py:47-58: log(1 + sum_{i,j: t_j < t_i} exp(s_j - s_i)), un-stabilised."""
    gap = logits[:, None, :] - logits[:, :, None]          # [n, i, j] = s_j - s_i
    lower = (targets[:, None, :] < targets[:, :, None]).to(logits.dtype)
    per_sample = torch.log(1 + (gap.exp() * lower).sum(2).sum(1))
    return per_sample.mean() if average else per_sample


def bce(logits, targets):
    """losses.py:19-22 with raw=True."""
    return F.binary_cross_entropy(torch.sigmoid(logits), targets)


def one_cycle_lr(step_index, max_steps, min_lr, max_lr):
    """ops/training.py:208-234 with linear annealing; step_index counts from 0."""
    knee = int(round(max_steps * 0.3))
    if step_index < knee:
        return min_lr + (step_index / knee) * (max_lr - min_lr)
    frac = (step_index - knee) / (max_steps - knee)
    return max_lr + frac * (min_lr / 1e3 - max_lr)


def make_adam(model, lr, weight_decay=0.0):
    """ops/training.py:9-12 -- Adam with amsgrad."""
    return torch.optim.Adam(model.parameters(), lr, weight_decay=weight_decay, amsgrad=True)


def train_step(model, optimizer, signal, labels, loss="lsep", accumulation_steps=1):
    """One iteration of classifiers.py:652-690 (forward, per-sample LSEP, mean, backward,
    optimizer step).  Returns (logits, per-sample loss)."""
    model.train()
    optimizer.zero_grad()
    logits = model(signal)["class_logits"]
    if loss == "lsep":
        per = lsep(logits, labels, average=False) / accumulation_steps
        per.mean().backward()
    else:
        per = bce(logits, labels)
        per.backward()
    optimizer.step()
    return logits.detach(), per.detach()


def state_dict_signature(model):
    return OrderedDict((k, (tuple(v.shape), str(v.dtype))) for k, v in model.state_dict().items())
