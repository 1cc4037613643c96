"""Device-side input pipeline: batched crop / chunk-permute / MixUp kernels fed by pinned double-buffered
uploads, bit-exact against (a) the fixture the imported reference produced with its own per-sample Compose and
SoundDataset (tests/golden/g13_input_pipeline.npz) and (b) this package's host transforms under the same seeds."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

from freesound_classification_amd.datasets.sound_dataset import SoundDataset  # noqa: E402
from freesound_classification_amd.ops.device_pipeline import DeviceInputPipeline  # noqa: E402
from freesound_classification_amd.ops.padding import make_collate_fn  # noqa: E402
from freesound_classification_amd.ops.transforms import (  # noqa: E402
    AudioFeatures, Compose, DropFields, MapLabels, MixUp, SampleLongAudio, ShuffleAudio, SyntheticAudio)

DEV = "cuda:0"


def seed_all(s):
    torch.manual_seed(s)
    np.random.seed(s)
    random.seed(s)


def _golden_dataset(g):
    class_map = {"c%02d" % i: i for i in range(80)}
    files = [str(f) for f in g["files"]]
    labels = [str(s).split(",") for s in g["labels_csv"]]
    return SoundDataset(files, labels, transform=Compose([SyntheticAudio(), MapLabels(class_map)])), class_map, files, labels


def test_device_pipeline_matches_reference_golden(golden):
    g = golden("g13_input_pipeline.npz")
    ds, _, files, _ = _golden_dataset(g)
    pipe = DeviceInputPipeline(ds, DEV, max_audio_length=int(g["max_length"]), p_shuffle=float(g["p_shuffle"]),
                               chunk_length=float(g["chunk_length"]), p_mixup=float(g["p_mixup"]))
    seed_all(int(g["seed"]))
    rows = []
    for idx in ([0, 1, 2, 3, 4], [5], list(range(6, len(files)))):          # any batching: draws are per sample, in order
        out = pipe.batch(idx)
        sig, lab = out["signal"].cpu().numpy(), out["labels"].cpu().numpy()
        assert sig.shape[0] == len(idx) and sig.shape[2] == 1
        for r, i in enumerate(idx):
            rows.append((i, sig[r, :, 0], lab[r]))
    mixed = 0
    for i, sig, lab in rows:
        want = g["audio.%d" % i]
        np.testing.assert_array_equal(sig[:want.size], want)               # bit-exact
        assert not sig[want.size:].any()                                   # collate padding
        np.testing.assert_array_equal(lab, g["labels.%d" % i])
        mixed += int(lab.sum() > len(set(str(g["labels_csv"][i]).split(","))))
    assert mixed >= 2          # the fixture exercises MixUp (labels OR-ed with a partner's)


def test_device_pipeline_equals_host_transforms_and_prefetch():
    """Same seeds -> the device batch equals collate(host Compose) bit for bit, through `iterate` (double-buffered
    uploads, next batch planned while the current one is consumed)."""
    names = ["c%02d" % i for i in range(80)]
    class_map = {n: i for i, n in enumerate(names)}
    rng = np.random.RandomState(4)
    sr = 16000
    lens = rng.randint(int(0.4 * sr), int(4.0 * sr), size=37)
    files = ["synthetic:%d:%d:%d" % (i, n, sr) for i, n in enumerate(lens)]
    labels = [[names[rng.randint(80)]] for _ in files]
    loader = [SyntheticAudio(), MapLabels(class_map)]
    clean = Compose([SyntheticAudio(), SampleLongAudio(2), MapLabels(class_map)])
    host_tf = Compose([SyntheticAudio(), SampleLongAudio(2), MapLabels(class_map), ShuffleAudio(chunk_length=0.5, p=0.5),
                       MixUp(p=0.4), AudioFeatures("mel_1024_512_64", verbose=False), DropFields(("audio", "filename", "sr"))])
    host_ds = SoundDataset(files, labels, transform=host_tf, clean_transform=clean)
    batches = [list(range(k, min(k + 8, len(files)))) for k in range(0, len(files), 8)]
    collate = make_collate_fn({"signal": 0.0})
    seed_all(99)
    want = [collate([host_ds[i] for i in b]) for b in batches]
    pipe = DeviceInputPipeline(SoundDataset(files, labels, transform=Compose(loader)), DEV, max_audio_length=2,
                               p_shuffle=0.5, chunk_length=0.5, p_mixup=0.4)
    seed_all(99)
    got = list(pipe.iterate(batches))
    assert len(got) == len(want)
    for w, gbatch in zip(want, got):
        assert torch.equal(gbatch["signal"].cpu(), w["signal"])
        assert torch.equal(gbatch["labels"].cpu(), w["labels"])
    # augmentations switched off (classifiers.py:836-837): draws continue, clips pass through cropped only
    pipe.switch_off_augmentations()
    host_tf.switch_off_augmentations()
    seed_all(5)
    w = collate([host_ds[i] for i in batches[0]])
    seed_all(5)
    assert torch.equal(pipe.batch(batches[0])["signal"].cpu(), w["signal"])


def test_device_pipeline_full_size_crop_property():
    """cfg-2 size: 128 clips of 12 .. 20 s @ 44.1 kHz cropped to 10 s on the device: every row is the slice
    [start, start + 441000) of its clip (start recovered from the seeded draws), output (128, 441000, 1)."""
    sr = 44100
    rng = np.random.RandomState(0)
    lens = rng.randint(12 * sr, 20 * sr, size=128)
    files = ["synthetic:%d:%d:%d" % (i, n, sr) for i, n in enumerate(lens)]
    labels = [["c00"]] * 128
    ds = SoundDataset(files, labels, transform=Compose([SyntheticAudio(), MapLabels({"c00": 0, "c01": 1})]))
    pipe = DeviceInputPipeline(ds, DEV, max_audio_length=10, p_shuffle=0.0, p_mixup=0.0)
    seed_all(3)
    out = pipe.batch(list(range(128)))
    assert out["signal"].shape == (128, 441000, 1)
    seed_all(3)
    starts = []
    for n in lens:
        starts.append(int(np.random.randint(0, n - 441000)))
        np.random.uniform()
        np.random.uniform()
    sig = out["signal"].cpu().numpy()[:, :, 0]
    for i in (0, 17, 127):
        clip = (0.1 * np.random.RandomState(i).standard_normal(int(lens[i]))).astype(np.float32)
        np.testing.assert_array_equal(sig[i], clip[starts[i]:starts[i] + 441000])
