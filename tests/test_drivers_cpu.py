"""Host-side pieces of the drivers: the mag-style experiment directory, stratified folds, CLI surface.  CPU only."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from freesound_classification_amd import experiment as mag  # noqa: E402
from freesound_classification_amd.ops import folds as pfolds  # noqa: E402


def _config():
    return {"network": {"num_conv_blocks": 5, "conv_base_depth": 64, "growth_rate": 2.0, "aggregation_type": "max"},
            "data": {"features": "mel_2048_1024_128", "_n_folds": 4, "_input_dim": 128, "p_mixup": 0.5},
            "train": {"batch_size": 64, "learning_rate": 0.01, "_save_every": 1},
            "label": "2d_cnn"}


def test_experiment_directory_layout_and_results(tmp_path):
    """Reference README.md:136-146: checkpoints, command, commit_hash, config.json, log, predictions, results.json,
    summaries; results registered with dotted keys nest (train_2d_cnn.py:365, :457)."""
    mag.use_custom_separator("-")
    with mag.Experiment(_config(), experiments_dir=str(tmp_path)) as exp:
        print("hello log")
        exp.register_directory("checkpoints")
        exp.register_directory("predictions")
        exp.register_directory("summaries")
        exp.register_result("fold0.metric", np.float64(0.5))
        exp.register_result("fold1.metric", 0.25)
        exp.register_result("metric", 0.4)
        assert exp.config.network.conv_base_depth == 64 and exp.config.data._n_folds == 4
        d = exp.directory
    assert sorted(os.listdir(d)) == ["checkpoints", "command", "commit_hash", "config.json", "log", "predictions",
                                     "results.json", "summaries"]
    assert "hello log" in open(os.path.join(d, "log")).read()
    assert json.load(open(os.path.join(d, "config.json"))) == _config()
    res = json.load(open(os.path.join(d, "results.json")))
    assert res == {"fold0": {"metric": 0.5}, "fold1": {"metric": 0.25}, "metric": 0.4}
    # identifier: underscore-prefixed parameters are left out, values joined with the custom separator
    name = os.path.basename(d)
    assert "64-cbd" in name.split("|") and "2d_cnn-l" in name.split("|") and "0.5-pm" in name.split("|")
    assert not any(part.endswith("-nf") or part.endswith("-id") or part.endswith("-se") for part in name.split("|"))
    # an existing experiment needs implicit_resuming (train_2d_cnn.py --resume)
    with pytest.raises(ValueError):
        mag.Experiment(_config(), experiments_dir=str(tmp_path))
    again = mag.Experiment(_config(), implicit_resuming=True, experiments_dir=str(tmp_path))
    assert again.results.to_dict()["fold0"]["metric"] == 0.5 and "fold1" in again.results.to_dict()
    assert again.checkpoints == os.path.join(d, "checkpoints")
    # re-open by path, read-only (predict_2d_cnn.py:66)
    ro = mag.Experiment(resume_from=d, write=False)
    assert ro.config.train.batch_size == 64 and ro.predictions.endswith("predictions")


def test_multilabel_stratified_folds_properties():
    rng = np.random.RandomState(0)
    n, c, k = 997, 80, 5
    y = (rng.uniform(size=(n, c)) < 0.015).astype(np.float32)
    y[np.arange(n), rng.randint(0, c, n)] = 1.0
    y[:7] = 0.0                                            # a few label-free rows
    y[:, 3] = 0.0
    y[[11, 500], 3] = 1.0                                  # a label with fewer examples than folds
    folds = pfolds.multilabel_stratified_folds(y, k, seed=42)
    assert folds.shape == (n,) and set(folds) == set(range(k))
    sizes = np.bincount(folds, minlength=k)
    assert sizes.max() - sizes.min() <= 0.01 * n          # labels are balanced first, fold sizes follow
    per_label = np.stack([y[folds == f].sum(0) for f in range(k)])          # (k, c)
    total = y.sum(0)
    # label counts per fold: far closer to the ideal share than a plain shuffled K-fold, the rare label (2 examples)
    # lands in two different folds
    dev = float(((per_label - total / k) ** 2).sum())
    plain_folds = np.random.RandomState(1).permutation(n) % k
    plain_dev = float(((np.stack([y[plain_folds == f].sum(0) for f in range(k)]) - total / k) ** 2).sum())
    assert dev < 0.5 * plain_dev, (dev, plain_dev)
    assert per_label[:, 3].max() == 1.0
    assert np.array_equal(folds, pfolds.multilabel_stratified_folds(y, k, seed=42))
    assert not np.array_equal(folds, pfolds.multilabel_stratified_folds(y, k, seed=43))
    # generator interface of ops/folds.py:14-24
    classmap = {"c%02d" % i: i for i in range(c)}
    names = [",".join("c%02d" % j for j in np.flatnonzero(row)) or "c00" for row in y]
    splits = list(pfolds.train_validation_data_stratified(np.arange(n), np.asarray(names, dtype=object), classmap, k, 42))
    assert len(splits) == k
    seen = np.concatenate([v for _, v in splits])
    assert sorted(seen) == list(range(n))
    for tr, va in splits:
        assert len(set(tr) & set(va)) == 0 and len(tr) + len(va) == n
    plain = list(pfolds.train_validation_data(np.arange(50), np.arange(50), 5, 1))
    assert sorted(np.concatenate([v for _, v in plain])) == list(range(50))


def test_train_driver_cli_has_the_reference_flags():
    """Every flag of the reference's train_2d_cnn.py:36-187 parses (values from README.md:200-214, command 3)."""
    import train_2d_cnn as drv
    args = drv.build_parser().parse_args(
        "--train_df data/train_curated.csv --train_data_dir data/train_curated/ --noisy_train_df data/train_noisy.csv "
        "--noisy_train_data_dir data/train_noisy/ --share_noisy --resume --test_data_dir data/test/ "
        "--sample_submission data/sample_submission.csv --classmap data/classmap.json --log_interval 10 --batch_size 50 "
        "--max_audio_length 15 --lr 0.003 --max_samples 100 --holdout_size 0.1 --epochs 150 "
        "--scheduler 1cycle_0.0001_0.005 --accumulation_steps 1 --save_every 20 --device cuda --aggregation_type max "
        "--num_conv_blocks 6 --start_deep_supervision_on 1 --conv_base_depth 100 --growth_rate 1.5 --weight_decay 0.0 "
        "--output_dropout 0.7 --p_mixup 0.5 --p_aug 0.0 --switch_off_augmentations_on 140 --features mel_2048_1024_128 "
        "--optimizer adam --folds 0 1 2 3 4 --n_folds 5 --kfold_seed 42 --num_workers 8 --label 2d_cnn".split())
    assert args.folds == [0, 1, 2, 3, 4] and args.conv_base_depth == 100 and args.share_noisy and args.resume
    assert drv.build_parser("1d_cnn").parse_args([]).label == "1d_cnn"
    import train_hierarchical_cnn  # noqa: F401  (imports the shared driver with the 1-d model class)


def test_default_conv_arithmetic_is_the_headline_one_and_the_drivers_can_change_it():
    """Round 6 ("ship what you measure"): with no environment variable the library's default convolution arithmetic is f16x6
    (arith 10, products to 2^-32: the reference's fp32 nn.Conv2d precision, classifiers.py:526-531) -- what bench.py's `value` is
    measured in; `--conv_arith` on both drivers selects the others; a descriptor's FSC_ARITH_DEFAULT means that default."""
    import ctypes as C
    import subprocess
    import sys
    import train_2d_cnn as drv
    from freesound_classification_amd import functional as F
    lib = F._lib.load()
    env = {k: v for k, v in os.environ.items() if k != "FSC_CONV_ARITH"}
    code = "from freesound_classification_amd import functional as F; print(F._lib.load().fsc_conv_default_arith())"
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=ROOT).stdout.strip() == "10"
    for name, want in (("f16x3", "3"), ("bf16x9", "9"), ("f32", "0")):
        out = subprocess.run([sys.executable, "-c", code], env=dict(env, FSC_CONV_ARITH=name), capture_output=True, text=True, cwd=ROOT)
        assert out.stdout.strip() == want, (name, out.stdout, out.stderr)
    if "FSC_CONV_ARITH" not in os.environ:
        assert lib.fsc_conv_default_arith() == 10 and F.get_conv_arith() == 10
    args = drv.build_parser().parse_args("--synthetic 8 --conv_arith bf16x9".split())
    assert args.conv_arith == "bf16x9" and drv.build_parser().parse_args([]).conv_arith is None
    src = open(os.path.join(ROOT, "predict_2d_cnn.py")).read()
    assert "--conv_arith" in src and "set_conv_arith(args.conv_arith)" in src
    # FSC_ARITH_DEFAULT resolves to the process default at the pre-split entry points too: the same tiling as arith = default
    d_def = F.ConvDesc(128, 100, 100, 64, 215, 3, 3, -1)
    d_exp = F.ConvDesc(128, 100, 100, 64, 215, 3, 3, lib.fsc_conv_default_arith())
    assert lib.fsc_conv_l16_supported(C.byref(d_def), 0) == lib.fsc_conv_l16_supported(C.byref(d_exp), 0) == 1
    assert lib.fsc_conv_l16_packed_floats(C.byref(d_def), 0) == lib.fsc_conv_l16_packed_floats(C.byref(d_exp), 0)
    assert lib.fsc_conv_l16_wgrad_workspace_bytes(C.byref(d_def)) == lib.fsc_conv_l16_wgrad_workspace_bytes(C.byref(d_exp))
    buf_a, buf_b = C.create_string_buffer(256), C.create_string_buffer(256)
    assert lib.fsc_conv_l16_plan_describe(C.byref(d_def), 0, buf_a, 256) == 0 and lib.fsc_conv_l16_plan_describe(C.byref(d_exp), 0, buf_b, 256) == 0
    assert buf_a.value == buf_b.value
