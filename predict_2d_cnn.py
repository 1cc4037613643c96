"""Inference driver: counterpart of the reference's predict_2d_cnn.py (:72-125) with the
length-grouped batching its README describes (README.md:37) wired in through `BucketingSampler`
(ops/padding.py:36-81, defined but never instantiated by the reference's own scripts).

Per fold: rebuild the model, load `checkpoints/fold_{k}/best_model.pth`, predict full-length clips
(no crop, pad-to-longest inside a batch); fold probabilities are averaged and written as a CSV with
the class columns of `get_class_names_from_classmap` plus `fname`.  Under `torch.distributed.run`
the batch list is sharded round-robin over ranks (every GPU holds all fold weights) and the
probabilities are gathered on rank 0.  Batch composition is the single-process sampler's, because
padding is not masked downstream.
"""
import argparse
import os
import random

import numpy as np
import torch
import torch.distributed as dist

from freesound_classification_amd import parallel
from freesound_classification_amd.datasets.sound_dataset import SoundDataset
from freesound_classification_amd.networks.classifiers import (
    HierarchicalCNNClassificationModel, TwoDimensionalCNNClassificationModel)
from freesound_classification_amd.ops.padding import BucketingSampler, make_collate_fn
from freesound_classification_amd.ops.transforms import (
    AudioFeatures, Compose, DropFields, LoadAudio, SyntheticAudio)
from freesound_classification_amd.ops.utils import get_class_names_from_classmap, load_json


def LoadedExperiment(directory):
    """An experiment directory written by train_2d_cnn.py (or by `mag`), opened read-only by path -- the reference's
    `Experiment(resume_from=args.experiment)` (predict_2d_cnn.py:66)."""
    from freesound_classification_amd.experiment import Experiment
    return Experiment(resume_from=directory, write=False)


class _WithLengths(SoundDataset):
    """SoundDataset plus the `lengths` attribute BucketingSampler needs (samples per clip)."""

    def __init__(self, files, lengths, transform):
        super().__init__(files, transform=transform)
        self.lengths = np.asarray(lengths)


def clip_lengths(files):
    out = []
    for f in files:
        if str(f).startswith("synthetic:"):
            out.append(int(str(f).split(":")[2]))
        else:
            from scipy.io import wavfile
            sr, data = wavfile.read(f, mmap=True)
            out.append(int(data.shape[0]))
    return out


def grouped_batches(dataset, bucket_seconds, max_batch_seconds, sr, seed):
    """Length-grouped batch list (identical on every rank: the sampler draws from `random`)."""
    random.seed(seed)
    top = int(np.max(dataset.lengths)) + 1
    step = int(bucket_seconds * sr)
    buckets = list(range(0, top + step, step))
    return [list(map(int, b)) for b in BucketingSampler(dataset, int(max_batch_seconds * sr), buckets)]


def load_fold_models(experiment, folds, device, model_cls):
    """All fold weight sets resident on the device at once (5 x 86 MB at cfg 2), in eval mode."""
    models = []
    for fold in folds:
        model = model_cls(experiment, device=device)
        model.load_best_model(fold)
        models.append(model.eval())
    return models


# fold models of an ensemble batch on this many streams (measured at cfg 5, 5 folds: 1 -> 1040, 2 -> 1124, 3 -> 1142, 4 -> 1118,
# 5 -> 816 clips/s in f16x6)
FOLD_STREAMS = int(os.environ.get("FSC_FOLD_STREAMS", "3"))
_FOLD_STREAMS = {}


def _fold_streams(device):
    key = str(device)
    if key not in _FOLD_STREAMS or len(_FOLD_STREAMS[key]) != FOLD_STREAMS:
        _FOLD_STREAMS[key] = [torch.cuda.Stream(device=device) for _ in range(FOLD_STREAMS)]
    return _FOLD_STREAMS[key]


def ensemble_batch(models, signal, scope=None):
    """Mean over the fold models of sigmoid(logits) for one device batch.  The front-end (STFT -> mel -> log) has no
    trained parameters, so it runs once and its output feeds every fold's conv stack.
    scope (functional.act_fold_scope()): the residual units run conv -> BatchNorm -> PReLU as one launch with calibrated operand
    scales where the arithmetic has scales (f16x6); the caller asks `scope.ok()` once the result is on its way to the host and
    calls again WITHOUT a scope if it is not (a scale was outgrown: never seen on the synthetic or the golden inputs)."""
    from freesound_classification_amd import functional as F
    import contextlib
    streams = _fold_streams(signal.device) if (FOLD_STREAMS > 1 and len(models) > 1 and signal.is_cuda) else None
    with torch.no_grad(), (scope if scope is not None else contextlib.nullcontext()):
        feats = models[0].features(signal)
        if streams is None:
            total = None
            for model in models:
                probs = F.sigmoid(model.forward_features(feats)["class_logits"])
                total = probs if total is None else total.add_(probs)
            return total.div_(len(models))
        # fold models on FOLD_STREAMS streams: the late blocks of one model (a few dozen work items per kernel) leave most of the
        # chip idle, which the other stream's kernels fill; the chip-filling early blocks just queue behind each other
        main = torch.cuda.current_stream(signal.device)
        parts = []
        if scope is not None:
            scope.prepare(signal, slots=64 * len(models))      # (the flag words exist, zeroed, before any side stream touches them)
        for s in streams:
            s.wait_stream(main)
        for i, model in enumerate(models):
            with torch.cuda.stream(streams[i % len(streams)]):
                parts.append(F.sigmoid(model.forward_features(feats)["class_logits"]))
        for s in streams:
            main.wait_stream(s)
            feats.record_stream(s)
        for p in parts:
            p.record_stream(main)
        total = parts[0]
        for p in parts[1:]:
            total = total.add_(p)
    return total.div_(len(models))


def ensemble_batch_checked(models, signal):
    """ensemble_batch on the folded route, verified: the result as a host array (the check rides on the copy's synchronisation)."""
    from freesound_classification_amd import functional as F
    # (ONE device-to-host copy and one synchronisation per batch: the overflow flags of the calibrated folds travel with the result)
    return F.eval_checked(lambda scope: ensemble_batch(models, signal, scope))


def ensemble_batches(models, signals, depth=2):
    """Generator over device batches: yields each batch's verified host probabilities, in order.  Batch i + 1 is enqueued BEFORE
    batch i's result is waited for (its probabilities and the overflow flags of its calibrated folds go to a pinned buffer in one
    asynchronous copy behind its kernels), so the GPU never idles on the host's per-batch synchronisation.  A batch whose flags
    say a calibrated scale was outgrown is recomputed on the two-pass route when its turn comes."""
    from collections import deque
    from freesound_classification_amd import functional as F
    pending = deque()

    def finish(item):
        x, scope, host, event, shape, n = item
        event.synchronize()
        if not scope.ok(host[n:]):
            F.EVAL_RECOMPUTES += 1
            return ensemble_batch(models, x).cpu()
        return host[:n].reshape(shape).clone()

    for x in signals:
        scope = F.act_fold_scope()
        out = ensemble_batch(models, x, scope)
        bad = scope.bad_mask()
        flat = out.reshape(-1) if bad is None else torch.cat([out.reshape(-1), bad])
        host = torch.empty(flat.numel(), dtype=torch.float32, pin_memory=True)
        host.copy_(flat, non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        pending.append((x, scope, host, event, tuple(out.shape), out.numel()))
        if len(pending) >= depth:
            yield finish(pending.popleft())
    while pending:
        yield finish(pending.popleft())


def predict_folds(experiment, folds, dataset, batches, collate, device, model_cls, models=None):
    """Mean over folds of sigmoid(logits); rows follow the dataset order.  Returns rank 0's array (the other ranks: zeros).
    (Reference predict_2d_cnn.py:72-125 loops folds outermost and reloads the data per fold; here every batch is
    uploaded once and goes through all resident fold models -- the same numbers, since eval-mode batches are independent.)
    N ranks: the length-grouped batches go round-robin (batch composition equals the single-process sampler's: padding is
    unmasked), every rank averages its folds locally, and ONE gather of (rows, n_classes) fp32 + the row indices brings
    them to rank 0 (parallel.gather_rows: tensors, no pickling)."""
    world, rank = parallel.world_size(), parallel.rank()
    mine = batches[rank::world]
    order = [i for b in mine for i in b]
    n_classes = experiment.config.data._n_classes
    if models is None:
        models = load_fold_models(experiment, folds, device, model_cls)
    chunks = []
    if order:
        loader = torch.utils.data.DataLoader(dataset, batch_sampler=mine, collate_fn=collate)
        for host in ensemble_batches(models, (sample["signal"].to(device) for sample in loader)):
            chunks.append(host.numpy())
    local = np.concatenate(chunks).astype(np.float32) if chunks else np.zeros((0, n_classes), np.float32)
    out = np.zeros((len(dataset), n_classes), np.float32)
    if world == 1:
        out[order] = local
        return out
    for idx, block in parallel.gather_rows(order, local, n_classes, device):
        out[idx] = block
    return out


def main():
    p = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--experiment", required=True)
    p.add_argument("--test_df", type=str)
    p.add_argument("--test_data_dir", type=str)
    p.add_argument("--classmap", type=str)
    p.add_argument("--synthetic", type=int, default=0, help="predict N synthetic clips of random lengths")
    p.add_argument("--synthetic_sr", type=int, default=44100)
    p.add_argument("--output_df", required=True)
    p.add_argument("--device", default="cuda")
    p.add_argument("--folds", type=int, nargs="+", default=None)
    p.add_argument("--model", choices=["2d", "1d"], default="2d")
    p.add_argument("--bucket_seconds", type=float, default=2.0)
    p.add_argument("--max_batch_seconds", type=float, default=1280.0, help="samples per batch = 128 x 10 s")
    p.add_argument("--seed", type=int, default=7)
    p.add_argument("--gpus", type=int, default=1, help="ranks to start on this node (one process per GPU, batches round-robin)")
    p.add_argument("--conv_arith", default=None, choices=("f16x6", "bf16x9", "f16x3", "f32", "bf16"),
                   help="arithmetic of the convolutions (default: the library's f16x6 = the reference's fp32 precision; "
                        "f16x3 is the opt-in fast mode with 22-bit products)")
    args = p.parse_args()
    if args.conv_arith is not None:
        from freesound_classification_amd import functional as F
        F.set_conv_arith(args.conv_arith)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import sys
        raise SystemExit(parallel.launch_ranks(__file__, sys.argv[1:], args.gpus))

    device = args.device
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl")
        device = "cuda:%d" % local
    experiment = LoadedExperiment(args.experiment)
    config = experiment.config
    if args.synthetic:
        rng = np.random.RandomState(args.seed)
        class_map = {"class_%02d" % i: i for i in range(config.data._n_classes)}
        lens = rng.randint(int(0.3 * args.synthetic_sr), int(30 * args.synthetic_sr), size=args.synthetic)
        files = ["synthetic:%d:%d:%d" % (i, n, args.synthetic_sr) for i, n in enumerate(lens)]
        loader_tf, sr = SyntheticAudio(), args.synthetic_sr
    else:
        import pandas as pd
        class_map = load_json(args.classmap)
        df = pd.read_csv(args.test_df)
        files = [os.path.join(args.test_data_dir, f) for f in df.fname.values]
        loader_tf, sr = LoadAudio(), 44100
    features = AudioFeatures(config.data.features, verbose=False)
    transform = Compose([loader_tf, features, DropFields(("audio", "filename", "sr"))])
    dataset = _WithLengths(files, clip_lengths(files), transform)
    batches = grouped_batches(dataset, args.bucket_seconds, args.max_batch_seconds, sr, args.seed)
    folds = args.folds
    if folds is None:
        folds = sorted(int(d.split("_")[1]) for d in os.listdir(experiment.checkpoints) if d.startswith("fold_"))
    model_cls = TwoDimensionalCNNClassificationModel if args.model == "2d" else HierarchicalCNNClassificationModel
    probs = predict_folds(experiment, folds, dataset, batches, make_collate_fn({"signal": features.padding_value}),
                          device, model_cls)
    if parallel.rank() == 0:
        import pandas as pd
        out = pd.DataFrame(probs, columns=get_class_names_from_classmap(class_map))
        out["fname"] = [os.path.basename(str(f)) for f in files]
        out.to_csv(args.output_df, index=False)
        print("wrote %d x %d predictions (%d folds, %d length-grouped batches) to %s" % (
            probs.shape[0], probs.shape[1], len(folds), len(batches), args.output_df))
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
