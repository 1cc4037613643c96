"""Training driver for the accelerated 2-d CNN: counterpart of the reference's train_2d_cnn.py
(CLI flags of train_2d_cnn.py:36-187 that concern the hot path, same config dictionary layout,
fit_validate -> predictions CSV -> out-of-fold lwlrap).

Data: either the reference's CSV layout (`--train_df` with columns fname, labels and
`--train_data_dir` with WAV files) or `--synthetic N` clips of seeded noise.  Launch under
`python -m torch.distributed.run --nproc-per-node N` for data-parallel training (one process per GPU).
`mag` is replaced by a minimal experiment directory object with the same attributes.
"""
import argparse
import json
import os
import random

import numpy as np
import torch
import torch.distributed as dist

from freesound_classification_amd import parallel
from freesound_classification_amd.datasets.sound_dataset import SoundDataset
from freesound_classification_amd.networks.classifiers import (
    HierarchicalCNNClassificationModel, TwoDimensionalCNNClassificationModel)
from freesound_classification_amd.ops.padding import make_collate_fn
from freesound_classification_amd.ops.transforms import (
    AudioFeatures, Compose, DropFields, LoadAudio, MapLabels, MixUp, SampleLongAudio, ShuffleAudio,
    SyntheticAudio)
from freesound_classification_amd.ops.utils import get_class_names_from_classmap, load_json, lwlrap


class AttrDict(dict):
    __getattr__ = dict.__getitem__


def to_attr(d):
    return AttrDict({k: to_attr(v) if isinstance(v, dict) else v for k, v in d.items()})


class Experiment:
    """Directory-backed stand-in for mag.Experiment: config.json, results.json, registered dirs."""

    def __init__(self, config, root="experiments"):
        self.config = to_attr(config)
        label = config.get("label") or "run"
        self.directory = os.path.join(root, label)
        os.makedirs(self.directory, exist_ok=True)
        with open(os.path.join(self.directory, "config.json"), "w") as f:
            json.dump(config, f, indent=2)
        self._results = {}

    def register_directory(self, name):
        path = os.path.join(self.directory, name)
        os.makedirs(path, exist_ok=True)
        setattr(self, name, path)

    def register_result(self, key, value):
        self._results[key] = value
        with open(os.path.join(self.directory, "results.json"), "w") as f:
            json.dump(self._results, f, indent=2)

    @property
    def results(self):
        return AttrDict(to_dict=lambda: dict(self._results))


def parse_args():
    p = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--train_df", type=str)
    p.add_argument("--train_data_dir", type=str)
    p.add_argument("--classmap", type=str)
    p.add_argument("--synthetic", type=int, default=0, help="use N synthetic clips instead of files")
    p.add_argument("--synthetic_seconds", type=float, default=2.0)
    p.add_argument("--synthetic_sr", type=int, default=16000)
    p.add_argument("--model", choices=["2d", "1d"], default="2d")
    p.add_argument("--device", default="cuda", choices=["cuda"])
    p.add_argument("--features", default="mel_1024_512_64")
    p.add_argument("--num_conv_blocks", type=int, default=3)
    p.add_argument("--conv_base_depth", type=int, default=32)
    p.add_argument("--growth_rate", type=float, default=2.0)
    p.add_argument("--start_deep_supervision_on", type=int, default=1)
    p.add_argument("--output_dropout", type=float, default=0.0)
    p.add_argument("--aggregation_type", default="max")
    p.add_argument("--loss", default="lsep", choices=["lsep", "bce"])
    p.add_argument("--optimizer", default="adam", choices=["adam", "momentum"])
    p.add_argument("--lr", type=float, default=1e-3)
    p.add_argument("--weight_decay", type=float, default=0.0)
    p.add_argument("--scheduler", default="1cycle_0.0001_0.005")
    p.add_argument("--batch_size", type=int, default=64)
    p.add_argument("--epochs", type=int, default=1)
    p.add_argument("--accumulation_steps", type=int, default=1)
    p.add_argument("--save_every", type=int, default=10)
    p.add_argument("--switch_off_augmentations_on", type=int, default=10 ** 6)
    p.add_argument("--p_mixup", type=float, default=0.0)
    p.add_argument("--max_audio_length", type=int, default=10)
    p.add_argument("--n_folds", type=int, default=2)
    p.add_argument("--folds", type=int, nargs="+", default=[0])
    p.add_argument("--kfold_seed", type=int, default=42)
    p.add_argument("--num_workers", type=int, default=0)
    p.add_argument("--log_interval", type=int, default=25)
    p.add_argument("--label", default="2d_cnn")
    return p.parse_args()


def main():
    args = parse_args()
    torch.manual_seed(42)
    np.random.seed(42)
    random.seed(42)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
        args.device = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))

    if args.synthetic:
        class_map = {"class_%02d" % i: i for i in range(80)}
        n = int(args.synthetic_seconds * args.synthetic_sr)
        rng = np.random.RandomState(0)
        files = ["synthetic:%d:%d:%d" % (i, n, args.synthetic_sr) for i in range(args.synthetic)]
        labels = [["class_%02d" % rng.randint(80)] for _ in files]
        loader_tf = SyntheticAudio()
    else:
        import pandas as pd
        class_map = load_json(args.classmap)
        df = pd.read_csv(args.train_df)
        files = [os.path.join(args.train_data_dir, f) for f in df.fname.values]
        labels = [item.split(",") for item in df.labels.values]
        loader_tf = LoadAudio()

    audio_transform = AudioFeatures(args.features)
    experiment = Experiment({
        "network": {"num_conv_blocks": args.num_conv_blocks, "start_deep_supervision_on": args.start_deep_supervision_on,
                    "conv_base_depth": args.conv_base_depth, "growth_rate": args.growth_rate,
                    "output_dropout": args.output_dropout, "aggregation_type": args.aggregation_type},
        "data": {"features": args.features, "_n_folds": args.n_folds, "_kfold_seed": args.kfold_seed,
                 "_input_dim": audio_transform.n_features, "_n_classes": len(class_map), "p_mixup": args.p_mixup,
                 "max_audio_length": args.max_audio_length},
        "train": {"accumulation_steps": args.accumulation_steps, "batch_size": args.batch_size,
                  "learning_rate": args.lr, "scheduler": args.scheduler, "optimizer": args.optimizer,
                  "epochs": args.epochs, "_save_every": args.save_every, "weight_decay": args.weight_decay,
                  "switch_off_augmentations_on": args.switch_off_augmentations_on},
        "label": args.label})
    experiment.register_directory("checkpoints")
    experiment.register_directory("predictions")

    order = np.random.RandomState(args.kfold_seed).permutation(len(files))
    fold_of = np.empty(len(files), int)
    fold_of[order] = np.arange(len(files)) % args.n_folds
    class_names = get_class_names_from_classmap(class_map)
    collate = make_collate_fn({"signal": audio_transform.padding_value})
    model_cls = TwoDimensionalCNNClassificationModel if args.model == "2d" else HierarchicalCNNClassificationModel
    scores = {}
    for fold in args.folds:
        train_idx = np.flatnonzero(fold_of != fold)
        valid_idx = np.flatnonzero(fold_of == fold)
        lo, hi = parallel.shard_range(len(train_idx))          # contiguous shard per rank
        train_idx = train_idx[lo:hi]

        def subset(idx):
            return [files[i] for i in idx], [labels[i] for i in idx]

        clean = Compose([loader_tf, SampleLongAudio(args.max_audio_length), MapLabels(class_map)])
        train_tf = Compose([loader_tf, SampleLongAudio(args.max_audio_length), MapLabels(class_map), ShuffleAudio(p=0.5),
                            MixUp(p=args.p_mixup), audio_transform, DropFields(("audio", "filename", "sr"))])
        valid_tf = Compose([loader_tf, MapLabels(class_map), audio_transform, DropFields(("audio", "filename", "sr"))])
        tf_files, tf_labels = subset(train_idx)
        vf_files, vf_labels = subset(valid_idx)
        train_loader = torch.utils.data.DataLoader(
            SoundDataset(tf_files, tf_labels, transform=train_tf, clean_transform=clean),
            shuffle=True, drop_last=True, batch_size=args.batch_size, collate_fn=collate,
            num_workers=args.num_workers, pin_memory=True)
        valid_loader = torch.utils.data.DataLoader(
            SoundDataset(vf_files, vf_labels, transform=valid_tf), shuffle=False, batch_size=args.batch_size,
            collate_fn=collate, num_workers=args.num_workers, pin_memory=True)
        model = model_cls(experiment, device=args.device, loss=args.loss)
        fold_scores = model.fit_validate(train_loader, valid_loader, epochs=args.epochs, fold=fold,
                                         log_interval=args.log_interval)
        scores[fold] = fold_scores
        if parallel.rank() == 0:
            experiment.register_result("fold{}.metric".format(fold), max(fold_scores))
            torch.save(model.state_dict(), os.path.join(experiment.checkpoints, "fold_{}".format(fold), "final_model.pth"))
            probs = model.predict(valid_loader)
            truth = np.stack([valid_loader.dataset[i]["labels"] for i in range(len(valid_idx))])
            print("fold", fold, "validation lwlrap of the final model: %.4f" % lwlrap(truth, probs))
            import pandas as pd
            out = pd.DataFrame(probs, columns=class_names)
            out["fname"] = vf_files
            out.to_csv(os.path.join(experiment.predictions, "val_preds_fold_{}.csv".format(fold)), index=False)
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
    return scores


if __name__ == "__main__":
    main()
