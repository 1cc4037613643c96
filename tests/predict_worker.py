"""Worker of tests/test_predict_dp_gpu.py: one rank of the sharded 5-fold-style inference (predict_2d_cnn.predict_folds,
reference predict_2d_cnn.py:72-125; SURVEY 8e "Inference (cfg5)").  Launched N times by torch.distributed.run with a `gloo`
rendezvous, every rank on the single visible GPU; rank 0 writes what predict_folds returned to <outdir>/world<N>.npy.
With N = 1 (no launcher) it is the single-process reference of the same call."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class NS(dict):
    __getattr__ = dict.__getitem__


FEATURES = "mel_1024_512_64"
SR = 16000


def make_experiment():
    return NS(config=NS(
        network=NS(num_conv_blocks=3, start_deep_supervision_on=1, conv_base_depth=32, growth_rate=1.5,
                   output_dropout=0.0, aggregation_type="max"),
        data=NS(features=FEATURES, _input_dim=64, _n_classes=80),
        train=NS(accumulation_steps=1, optimizer="adam", learning_rate=1e-3, weight_decay=0.0,
                 scheduler="1cycle_0.0001_0.005")))


def make_models(n_folds=2):
    from freesound_classification_amd.networks.classifiers import TwoDimensionalCNNClassificationModel
    models = []
    for fold in range(n_folds):
        torch.manual_seed(300 + fold)                      # (every rank builds the same fold weights)
        m = TwoDimensionalCNNClassificationModel(make_experiment(), device="cuda:0")
        g = torch.Generator().manual_seed(900 + fold)
        for mod in m.modules():
            if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
                mod.running_mean.copy_(0.1 * torch.randn(mod.running_mean.shape, generator=g))
                mod.running_var.copy_(0.5 + torch.rand(mod.running_var.shape, generator=g))
        models.append(m.eval())
    return models


def make_dataset(n_clips, seed=11):
    import predict_2d_cnn as drv
    from freesound_classification_amd.ops.transforms import AudioFeatures, Compose, DropFields, SyntheticAudio
    rng = np.random.RandomState(seed)
    lens = rng.randint(int(0.4 * SR), int(3.0 * SR), size=n_clips)
    files = ["synthetic:%d:%d:%d" % (i, n, SR) for i, n in enumerate(lens)]
    features = AudioFeatures(FEATURES, verbose=False)
    transform = Compose([SyntheticAudio(), features, DropFields(("audio", "filename", "sr"))])
    return drv._WithLengths(files, drv.clip_lengths(files), transform), features


def run(n_clips, max_batch_seconds, bucket_seconds=0.5, arith=None, fold=True):
    import predict_2d_cnn as drv
    from freesound_classification_amd import functional as F
    from freesound_classification_amd.ops.padding import make_collate_fn
    F.set_conv_arith(arith)
    F.EVAL_ACT_FOLD = bool(fold)
    F.forget_packed_weights()                              # (a fresh calibration history, as in a fresh process)
    dataset, features = make_dataset(n_clips)
    batches = drv.grouped_batches(dataset, bucket_seconds, max_batch_seconds, SR, seed=3)
    probs = drv.predict_folds(make_experiment(), [0, 1], dataset, batches, make_collate_fn({"signal": features.padding_value}),
                              "cuda:0", None, models=make_models())
    return probs, batches


def main():
    outdir, n_clips, max_batch_seconds, bucket_seconds = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4])
    arith, fold = (None if sys.argv[5] == "default" else sys.argv[5]), sys.argv[6] == "1"
    dist.init_process_group("gloo")
    probs, batches = run(n_clips, max_batch_seconds, bucket_seconds, arith, fold)
    if dist.get_rank() == 0:
        np.save(os.path.join(outdir, "world%d.npy" % dist.get_world_size()), probs)
        np.save(os.path.join(outdir, "nbatches%d.npy" % dist.get_world_size()), np.asarray([len(batches)]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
