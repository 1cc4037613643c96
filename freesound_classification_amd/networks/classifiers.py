"""Counterpart of the reference's networks/classifiers.py for the accelerated path.

`TwoDimensionalCNNClassificationModel` and `HierarchicalCNNClassificationModel` keep the
reference's constructor (`experiment`, `device`), methods (forward / fit_validate /
train_epoch / evaluate / validation / predict / make_optimizer / load_best_model) and --
because the parameter holders are registered in the same order with the same torch module
types -- the same state-dict keys, shapes and default initialisation draws
(networks/classifiers.py:107-217, 483-892).  All device arithmetic runs in libfsc_hip.so.

Additions the reference does not have: `loss` selection ("lsep" | "bce"; cfg 1 of the
benchmark names BCE although the reference hard-codes LSEP), and data-parallel training when
`torch.distributed` is initialised (one process per GPU, bucketed RCCL all-reduce of the
gradients overlapped with backward, see `..parallel`).
"""
import gc
import os
from collections import deque

import numpy as np
import torch
import torch.nn as nn

from .. import functional as F
from .. import parallel
from ..ops.training import OPTIMIZERS, make_scheduler, make_step
from ..ops.utils import is_mel, is_stft, lwlrap, make_mel_filterbanks, parse_features
from .losses import binary_cross_entropy, lsep_loss

try:  # logging only; absent in this image
    from tensorboardX import SummaryWriter
except ImportError:  # pragma: no cover
    class SummaryWriter:
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

        add_image = add_histogram = add_scalar

try:
    from tqdm import tqdm
except ImportError:  # pragma: no cover
    tqdm = None


class _ResidualHolder(nn.Module):
    """Parameter holder for the bottleneck unit conv1x1-BN-PReLU-conv3-BN-PReLU-conv1x1-BN-(+x)-PReLU
    (reference ResnetBlock :37-69 and ResnetBlock2d :72-104).  The arithmetic lives in
    functional.ConvBlockFn; registration order fixes state-dict keys and init RNG order."""

    def __init__(self, depth, conv, bn):
        super().__init__()
        self.conv1 = conv(depth, depth, kernel_size=1)
        self.bn1 = bn(depth)
        self.conv2 = conv(depth, depth, kernel_size=3, padding=1)
        self.bn2 = bn(depth)
        self.conv3 = conv(depth, depth, kernel_size=1)
        self.bn3 = bn(depth)
        self.prelu1 = nn.PReLU(depth)
        self.prelu2 = nn.PReLU(depth)
        self.prelu3 = nn.PReLU(depth)

    def forward(self, x):
        raise RuntimeError("residual units run inside functional.conv_block")


class ResnetBlock(_ResidualHolder):
    def __init__(self, depth):
        super().__init__(depth, nn.Conv1d, nn.BatchNorm1d)


class ResnetBlock2d(_ResidualHolder):
    def __init__(self, depth):
        super().__init__(depth, nn.Conv2d, nn.BatchNorm2d)


class _TaggingModel(nn.Module):
    dims = 2

    def __init__(self, experiment, device="cuda", loss="lsep", sync_bn=None):
        super().__init__()
        self.device = device
        self.experiment = experiment
        self.config = experiment.config
        self.loss_name = loss
        net, data = self.config.network, self.config.data
        self.aggregation = getattr(net, "aggregation_type", "max")
        if self.aggregation not in ("max", "rnn"):
            raise ValueError("aggregation_type must be 'max' or 'rnn', got %r" % (self.aggregation,))
        if torch.device(device).type != "cuda":
            raise F._lib.FscError("the accelerated model needs device='cuda' (an MI355X under ROCm); "
                                  "there is no CPU fallback")
        self._bands = None
        if is_mel(data.features):
            self.filterbanks = torch.from_numpy(make_mel_filterbanks(data.features)).to(self.device)

        conv, bn, pool = (nn.Conv2d, nn.BatchNorm2d, nn.MaxPool2d) if self.dims == 2 else \
                         (nn.Conv1d, nn.BatchNorm1d, nn.MaxPool1d)
        res = ResnetBlock2d if self.dims == 2 else ResnetBlock
        self.conv_modules = nn.ModuleList()
        self.rnns = nn.ModuleList()
        total_depth = 0
        depth = None
        for k in range(net.num_conv_blocks):
            input_size = (2 if self.dims == 2 else data._input_dim) if not k else depth
            depth = int(net.growth_rate ** k * net.conv_base_depth)
            if k >= net.start_deep_supervision_on:
                if self.aggregation == "max":
                    total_depth += depth
                else:                              # (created BEFORE the block's conv modules, like the reference: :514-522)
                    rnn_size = 128
                    total_depth += rnn_size * 2
                    self.rnns.append(nn.Sequential(
                        nn.LayerNorm((depth,)), nn.GRU(depth, rnn_size, batch_first=True, bidirectional=True)))
            self.conv_modules.append(nn.Sequential(
                bn(input_size), conv(input_size, depth, kernel_size=3, padding=1),
                pool(kernel_size=2, stride=2), bn(depth), nn.PReLU(depth), res(depth)))
        self.output_transform = nn.Sequential(
            nn.BatchNorm1d(total_depth), nn.Linear(total_depth, total_depth),
            nn.BatchNorm1d(total_depth), nn.PReLU(total_depth),
            nn.Dropout(p=net.output_dropout), nn.Linear(total_depth, data._n_classes))
        self.to(self.device)
        self._reducer = None
        # cross-replica BatchNorm statistics under data parallelism (off: each replica normalises its own shard, like
        # N independent copies of the reference at the per-GPU batch); `config.train.sync_bn` or the constructor argument
        if sync_bn is None:
            try:                                   # (mag configs and plain attr-dicts raise different errors)
                sync_bn = self.config.train.sync_bn
            except (AttributeError, KeyError):
                sync_bn = False
        self.sync_bn = bool(sync_bn)
        self._bn_sync = None
        self._dropout_state = F.DropoutState()
        self._froze_gc = False

    # ------------------------------------------------------------------ forward
    def _front_end(self, signal):
        features = self.config.data.features
        wave = signal.squeeze(-1)
        two_d = self.dims == 2
        if is_mel(features):
            _, (n_fft, hop, _n_mel) = parse_features(features)
            if self._bands is None or self._bands.weights.device != wave.device:
                self._bands = F.MelBands(self.filterbanks.cpu().numpy(), wave.device)
            x = F.frontend_logmel(wave, n_fft, hop, self._bands, freq_channel=two_d)
            return x if two_d else x.squeeze(1)
        if is_stft(features):
            _, (n_fft, hop) = parse_features(features)
            return F.frontend_stft(wave, n_fft, hop, apply_log=True, freq_channel=two_d)
        raise NotImplementedError("features=%r: raw waveforms are not on the accelerated path" % features)

    def _on_device(self):
        """libfsc_hip launches on the CURRENT HIP device / stream: make it the model's for the duration of a call
        (a model built with device="cuda:1" in a process whose current device is 0)."""
        dev = torch.device(self.device)
        return torch.cuda.device(dev if dev.index is not None else torch.cuda.current_device())

    def forward(self, signal):
        with self._on_device():
            return self._forward(signal)

    def _forward(self, signal):
        return self._forward_features(self._front_end(signal))

    def features(self, signal):
        """Front-end only: (N, T, 1) waveforms -> the spectrogram tensor the conv blocks consume.  An ensemble of
        fold models shares it (same descriptor, same filterbank): see predict_2d_cnn.predict_folds."""
        with self._on_device():
            return self._front_end(signal)

    def forward_features(self, h):
        """Conv blocks + heads + classifier on a precomputed front-end output."""
        with self._on_device():
            return self._forward_features(h)

    def _forward_features(self, h):
        token = None
        if self.training and torch.is_grad_enabled():
            token = F.prepack_begin((id(self), tuple(h.shape)))     # all L16 weight fragments of the step in a few launches
        if self.training:
            F.counters_begin()                                       # every BatchNorm's num_batches_tracked in one launch
        try:
            return self._forward_blocks(h)
        finally:
            F.counters_flush()
            F.prepack_end(token)

    def _forward_blocks(self, h):
        if self.dims == 1:
            h = h.unsqueeze(2)              # (N, C, 1, L): the 1-d model is the H == 1 case
        start = self.config.network.start_deep_supervision_on
        ph = 2 if self.dims == 2 else 1
        feats = []
        for k, mods in enumerate(self.conv_modules):
            use_max = k >= start and self.aggregation == "max"
            h, feat = F.conv_block(h, mods, self.training, use_max, ph, self._bn_sync, next_bn=k + 1 < len(self.conv_modules))
            if k >= start and self.aggregation == "rnn":
                feat = F.rnn_head(h, self.rnns[k - start])
            if feat is not None:
                feats.append(feat)
        feats = F.cat_features(feats)
        ot = self.output_transform
        z = F.bn_act(feats, ot[0], None, self.training, self._bn_sync)
        z = F.linear(z, ot[1].weight, ot[1].bias)
        z = F.bn_act(z, ot[2], ot[3], self.training, self._bn_sync)
        z = F.dropout(z, ot[4].p, self.training, self._dropout_state)
        logits = F.linear(z, ot[5].weight, ot[5].bias)
        return dict(class_logits=logits)

    # ------------------------------------------------------------------ summaries
    def add_scalar_summaries(self, loss, metric, writer, global_step):
        writer.add_scalar("loss", loss, global_step)
        writer.add_scalar("metric", metric, global_step)

    def add_histogram_summaries(self, losses, writer, global_step):
        writer.add_histogram("losses", np.array(losses), global_step=global_step)

    def add_image_summaries(self, signal, global_step, writer, to_plot=8):
        pass   # the reference plots the raw waveform grid via torchvision; logging only

    # ------------------------------------------------------------------ training
    def _per_sample_loss(self, class_logits, labels):
        if self.loss_name == "lsep":
            return lsep_loss(class_logits, labels, average=False)
        if self.loss_name == "bce":
            return binary_cross_entropy(class_logits, labels)
        raise ValueError("unknown loss %r" % self.loss_name)

    def training_step(self, signal, labels, step_optimizer=True):
        """Forward, loss, backward, (all-reduce,) optimizer step on one device batch.
        Returns (class_logits, per-sample losses or scalar loss).  No host synchronisation."""
        acc = self.config.train.accumulation_steps
        with self._on_device():
            outputs = self(signal)
            class_logits = outputs["class_logits"]
            per = self._per_sample_loss(class_logits, labels)
            loss = F.mean(per, 1.0 / acc) if per.dim() else F.mean(per.reshape(1), 1.0 / acc)
            if self._reducer is not None:
                self._reducer.prepare(sync=step_optimizer)
            loss.backward()
            if step_optimizer:
                if self._reducer is not None:
                    self._reducer.finish()
                self.optimizer.step()
                self.optimizer.zero_grad()
        return class_logits, per, loss

    def train_epoch(self, train_loader, epoch, log_interval, write_summary=True):
        with self._on_device():
            return self._train_epoch(train_loader, epoch, log_interval, write_summary)

    def _train_epoch(self, train_loader, epoch, log_interval, write_summary=True):
        self.train()
        print("\n" + " " * 10 + "****** Epoch {epoch} ******\n".format(epoch=epoch))
        training_losses = []
        history = deque(maxlen=30)
        self.optimizer.zero_grad()
        acc = self.config.train.accumulation_steps
        pb = tqdm(total=len(train_loader), ncols=80) if tqdm is not None else None
        for batch_idx, sample in enumerate(train_loader):
            self.global_step += 1
            make_step(self.scheduler, step=self.global_step)
            signal = sample["signal"].to(self.device, non_blocking=True)
            labels = sample["labels"].to(self.device, non_blocking=True).float()
            # the reference steps when batch_idx % accumulation_steps == 0 (so also on batch 0)
            class_logits, per, loss = self.training_step(signal, labels, batch_idx % acc == 0)
            probs = F.sigmoid(class_logits)
            # one host synchronisation per step for the running metric (the reference has three)
            per_host = per.detach().reshape(-1).cpu().numpy() / acc
            training_losses.extend(per_host)
            metric = lwlrap(labels.cpu().numpy(), probs.cpu().numpy())
            history.append(metric)
            loss_value = float(per_host.mean())
            if pb is not None:
                pb.update()
                pb.set_description("Loss: {:.4f}, Metric: {:.4f}".format(loss_value, np.mean(history)))
            if write_summary and batch_idx % log_interval == 0:
                self.add_scalar_summaries(loss_value, metric, self.train_writer, self.global_step)
        if pb is not None:
            pb.close()
        if write_summary:
            self.add_histogram_summaries(training_losses, self.train_writer, self.global_step)

    def evaluate(self, loader, verbose=False, write_summary=False, epoch=None):
        with self._on_device():
            return self._evaluate(loader, verbose, write_summary, epoch)

    def _evaluate(self, loader, verbose=False, write_summary=False, epoch=None):
        self.eval()
        valid_loss = 0.0
        all_class_probs, all_labels = [], []
        with torch.no_grad():
            for sample in loader:
                signal = sample["signal"].to(self.device)
                labels = sample["labels"].to(self.device).float()
                class_logits = self._eval_logits(signal)
                loss = lsep_loss(class_logits, labels).item()
                valid_loss += loss * len(labels) / len(loader.dataset)
                all_class_probs.extend(F.sigmoid(class_logits).cpu().numpy())
                all_labels.extend(labels.cpu().numpy())
        all_class_probs = np.asarray(all_class_probs)
        all_labels = np.asarray(all_labels)
        metric = lwlrap(all_labels, all_class_probs)
        self.last_valid_loss = valid_loss
        if write_summary:
            self.add_scalar_summaries(valid_loss, metric, writer=self.valid_writer, global_step=self.global_step)
        if verbose:
            print("\nValidation loss: {:.4f}".format(valid_loss))
            print("Validation metric: {:.4f}".format(metric))
        return metric

    def _eval_logits(self, signal):
        """Eval-mode logits of one batch on the folded inference route (conv -> BatchNorm -> PReLU of the residual units as one
        launch), VERIFIED: the model opens the fold scope and checks it itself (functional.eval_checked: the
        overflow flags of the calibrated scales ride on the logits' device-to-host copy; a batch that outgrew a calibration is
        recomputed on the two-pass route), so predict() / evaluate() -- the reference's own API, classifiers.py:709-797 -- get
        the fold in every arithmetic without the caller knowing about it.  Returns a device tensor."""
        with torch.no_grad():
            host = F.eval_checked(lambda scope: self(signal)["class_logits"])
        return host.to(signal.device)

    def validation(self, valid_loader, epoch):
        return self.evaluate(valid_loader, verbose=True, write_summary=True, epoch=epoch)

    def predict(self, loader, n_tta=1):
        with self._on_device():
            return self._predict(loader, n_tta)

    def _predict(self, loader, n_tta=1):
        self.eval()
        all_class_probs = []
        for _ in range(n_tta):
            tta_probs = []
            with torch.no_grad():
                for sample in loader:
                    signal = sample["signal"].to(self.device)
                    tta_probs.extend(F.eval_checked(lambda scope: F.sigmoid(self(signal)["class_logits"])).numpy())
            all_class_probs.append(np.array(tta_probs))
        return np.mean(all_class_probs, 0)

    def fit_validate(self, train_loader, valid_loader, epochs, fold, log_interval=25):
        self.experiment.register_directory("summaries")
        fold_dir = "fold_{}".format(fold)
        self.train_writer = SummaryWriter(log_dir=os.path.join(self.experiment.summaries, fold_dir, "train"))
        self.valid_writer = SummaryWriter(log_dir=os.path.join(self.experiment.summaries, fold_dir, "valid"))
        os.makedirs(os.path.join(self.experiment.checkpoints, fold_dir), exist_ok=True)
        self.global_step = 0
        self.make_optimizer(max_steps=len(train_loader) * epochs)
        scores = []
        best_score = 0
        for epoch in range(epochs):
            make_step(self.scheduler, epoch=epoch)
            if epoch == self.config.train.switch_off_augmentations_on:
                train_loader.dataset.transform.switch_off_augmentations()
            self.train_epoch(train_loader, epoch, log_interval, write_summary=True)
            validation_score = self.validation(valid_loader, epoch)
            scores.append(validation_score)
            if parallel.rank() != 0:
                continue
            ckpt = os.path.join(self.experiment.checkpoints, fold_dir)
            if epoch % self.config.train._save_every == 0:
                print("\nSaving model on epoch", epoch)
                torch.save(self.state_dict(), os.path.join(ckpt, "model_on_epoch_{}.pth".format(epoch)))
            if validation_score > best_score:
                torch.save(self.state_dict(), os.path.join(ckpt, "best_model.pth"))
                best_score = validation_score
        return scores

    def close(self):
        """End of a training run with this model: detach the gradient reducer (its autograd hooks hold it and it holds the
        parameters -- a cycle; the buckets are 86 MB at cfg 2), stop redirecting weight gradients, and return the objects
        make_optimizer parked in the garbage collector's permanent generation.  The drivers call it once per fold."""
        if self._reducer is not None:
            if F.GRAD_OUT == self._reducer.grad_view:
                F.GRAD_OUT = None
            self._reducer.remove()
            self._reducer = None
        self._bn_sync = None
        F.prepack_forget(id(self))
        F.forget_packed_weights()
        if self._froze_gc:
            gc.unfreeze()
            self._froze_gc = False

    def make_optimizer(self, max_steps):
        train = self.config.train
        if self._reducer is not None:              # (a second call on the same model: drop the previous reducer's hooks first)
            self.close()
        self.optimizer = OPTIMIZERS[train.optimizer](
            self.parameters(), train.learning_rate, weight_decay=train.weight_decay)
        self.scheduler = make_scheduler(train.scheduler, max_steps=max_steps)(self.optimizer)
        # FSC_FORCE_DP=1 exercises the data-parallel path in a 1-rank process group (device test)
        if parallel.world_size() > 1 or (parallel.initialized() and os.environ.get("FSC_FORCE_DP") == "1"):
            parallel.broadcast_module(self)
            self._reducer = parallel.BucketedGradReducer(list(self.parameters()))
            F.GRAD_OUT = self._reducer.grad_view       # weight gradients are written straight into the buckets
            self.optimizer.grad_scale = 1.0 / parallel.world_size()
            if self.sync_bn:
                self._bn_sync = parallel.SyncBN()
            # every replica is seeded alike (same initial weights): give each its own stretch of the counter-based dropout
            # stream, or row i of every rank would draw the same mask each step
            self._dropout_state.offset = parallel.rank() << 40
        # The interpreter's full garbage collections walk every object imported so far (torch alone: ~100 ms) and the training
        # loop triggers one every few steps -- in the data-parallel path (hooks, bucket views) as early as the sixth step, a
        # 100 ms hole in a 37 ms step.  Everything alive now is long-lived: park it in the permanent generation.
        # Undone by close() (fit_validate's callers: once per fold).
        gc.collect()
        gc.freeze()
        self._froze_gc = True

    def load_best_model(self, fold):
        path = os.path.join(self.experiment.checkpoints, "fold_{}".format(fold), "best_model.pth")
        self.load_state_dict(torch.load(path, map_location=self.device))

    def load_state_dict(self, *args, **kwargs):
        F.forget_packed_weights()                  # (copy_ bumps the versions too; explicit so that no cache outlives a checkpoint)
        return super().load_state_dict(*args, **kwargs)

    def train(self, mode=True):
        if mode:
            F.forget_packed_weights()              # inference caches do not survive a return to training
        return super().train(mode)


class TwoDimensionalCNNClassificationModel(_TaggingModel):
    """Reference networks/classifiers.py:483-892 (log-mel + frequency channel, 2-d blocks)."""
    dims = 2


class HierarchicalCNNClassificationModel(_TaggingModel):
    """Reference networks/classifiers.py:107-480 (spectrogram bins as channels, 1-d blocks).
    The reference's 1-d training loop takes the scalar-mean LSEP of `logits.squeeze()` (:268-275); here both models share
    the per-sample LSEP followed by the mean (`_per_sample_loss`): the same number for batches of two or more clips.  At
    batch 1 the reference's `squeeze()` drops the batch axis and its loss indexes a 1-d tensor -- not reproduced."""
    dims = 1
