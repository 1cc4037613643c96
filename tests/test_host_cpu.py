"""CPU-side checks of the product package: host logic against the golden vectors, the C-ABI
library (loads, exports every symbol include/fsc_hip.h declares), and the loud failure when the
accelerated path is asked to run without a GPU.  No kernel is launched here."""
import ctypes
import os
import random
import re

import numpy as np
import pytest
import torch

import freesound_classification_amd as pkg
from freesound_classification_amd import _lib
from freesound_classification_amd.datasets.sound_dataset import SoundDataset
from freesound_classification_amd.ops import audio as paudio
from freesound_classification_amd.ops import padding as ppad
from freesound_classification_amd.ops import training as ptrain
from freesound_classification_amd.ops import transforms as ptf
from freesound_classification_amd.ops import utils as putils

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "fsc_hip.h")).read()
    declared = set(re.findall(r"\b(fsc_[a-z0-9_]+)\s*\(", header))
    declared -= {"fsc_stream_t"}
    assert len(declared) >= 35
    lib = ctypes.CDLL(pkg.library_path())
    for name in sorted(declared):
        assert hasattr(lib, name), "libfsc_hip.so does not export %s" % name
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())
    assert pkg.load_library().fsc_version() >= 100
    assert pkg.load_library().fsc_last_error_string() is not None


def test_no_cpu_fallback():
    from freesound_classification_amd import functional as F
    from freesound_classification_amd.networks.classifiers import TwoDimensionalCNNClassificationModel
    from freesound_classification_amd.networks.losses import lsep_loss

    class NS(dict):
        __getattr__ = dict.__getitem__

    exp = NS(config=NS(network=NS(num_conv_blocks=1, start_deep_supervision_on=0, conv_base_depth=4, growth_rate=2,
                                  output_dropout=0.0, aggregation_type="max"),
                       data=NS(features="mel_1024_512_64", _input_dim=64, _n_classes=80),
                       train=NS(accumulation_steps=1)))
    with pytest.raises(_lib.FscError):
        TwoDimensionalCNNClassificationModel(exp, device="cpu")
    with pytest.raises(_lib.FscError):
        lsep_loss(torch.zeros(2, 80), torch.zeros(2, 80))
    with pytest.raises(_lib.FscError):
        F.frontend_stft(torch.zeros(1, 4000), 256, 128, True)
    with pytest.raises(_lib.FscError):
        putils.compute_torch_stft(torch.zeros(1, 4000), "stft_256_128")


def test_argument_errors_surface_as_text():
    lib = pkg.load_library()
    rc = lib.fsc_frontend_tables_init(None, 100, None)
    assert rc != 0
    assert b"power of two" in lib.fsc_last_error_string()
    d = _lib.ConvDesc(1, 3, 4, 5, 5, 5, 5)        # 5x5 kernels are not on the path
    assert lib.fsc_conv_packed_floats(ctypes.byref(d), 0) == 0


def test_conv_planner_covers_benchmark_shapes():
    lib = pkg.load_library()
    depths = [int(1.5 ** k * 100) for k in range(6)]
    h, w, cin = 128, 431, 2
    for dpt in depths:
        for (ci, co, hh, ww, kh, kw) in [(cin, dpt, h, w, 3, 3), (dpt, dpt, h // 2, w // 2, 1, 1),
                                         (dpt, dpt, h // 2, w // 2, 3, 3)]:
            d = _lib.ConvDesc(128, ci, co, hh, ww, kh, kw)
            for mode in (0, 1, 2):
                buf = ctypes.create_string_buffer(256)
                assert lib.fsc_conv_plan_describe(ctypes.byref(d), mode, buf, 256) == 0
                assert b"kernel<" in buf.value
            assert lib.fsc_conv_packed_floats(ctypes.byref(d), 0) >= ci * co * kh * kw
            assert lib.fsc_conv_wgrad_workspace_bytes(ctypes.byref(d)) >= 4 * ci * co * kh * kw
        h, w, cin = h // 2, w // 2, dpt


def test_mel_filterbank_matches_fixture(golden):
    for desc, ref in golden("g2_filterbanks.npz").items():
        np.testing.assert_array_equal(putils.make_mel_filterbanks(desc), ref)
    assert putils.is_mel("mel_1_2_3") and putils.is_stft("stft_256_128") and not putils.is_mel("raw")


def test_mel_bands_reproduce_dense_product(golden):
    from freesound_classification_amd.functional import MelBands
    fb = golden("g2_filterbanks.npz")["mel_2048_1024_128"]
    bands = MelBands(fb, "cpu")
    dense = np.zeros_like(fb)
    st, ln, w = bands.start.numpy(), bands.length.numpy(), bands.weights.numpy()
    for m in range(fb.shape[0]):
        dense[m, st[m]:st[m] + ln[m]] = w[:ln[m], m]
    np.testing.assert_array_equal(dense, fb)
    assert bands.max_band <= 64


def test_lwlrap_matches_reference_value(golden):
    g = golden("g10_lwlrap.npz")
    assert abs(putils.lwlrap(g["truth"], g["scores"]) - float(g["value"])) < 1e-12


@pytest.mark.parametrize("case", ["long_first", "short_first", "equal"])
def test_mixup_host_bit_exact(golden, case):
    g = golden("g7_mixup.npz")
    np.random.seed(70)
    random.seed(70)
    mixed, labels = paudio.mix_audio_and_labels(g[case + ".a"].copy(), g[case + ".b"].copy(),
                                                g[case + ".ya"], g[case + ".yb"])
    np.testing.assert_array_equal(mixed, g[case + ".mixed"])
    np.testing.assert_array_equal(labels, g[case + ".labels"])


def test_collate_bit_exact(golden):
    g = golden("g8_collate.npz")
    batch = [dict(signal=g["collate.in%d" % i].copy(), labels=g["collate.labels"][i], is_noisy=np.float64(i % 2))
             for i in range(4)]
    out = ppad.make_collate_fn({"signal": 0.0})(batch)
    np.testing.assert_array_equal(out["signal"].numpy(), g["collate.signal"])
    np.testing.assert_array_equal(out["labels"].numpy(), g["collate.labels"])
    np.testing.assert_array_equal(out["is_noisy"].numpy(), g["collate.is_noisy"])
    assert [str(out[k].dtype) for k in ("signal", "labels", "is_noisy")] == list(g["collate.dtypes"])
    edge = ppad.make_collate_fn({"signal": "edge"})([dict(signal=np.arange(3.0)[:, None]),
                                                    dict(signal=np.arange(5.0)[:, None])])
    assert edge["signal"][0, :, 0].tolist() == [0, 1, 2, 2, 2]


@pytest.mark.parametrize("case", ["small", "large"])
def test_bucketing_sampler_bit_exact(golden, case):
    g = golden("g8_bucketing.json")[case]

    class DS:
        lengths = np.array(g["lengths"])

    random.seed(g["seed"])
    s = ppad.BucketingSampler(DS(), g["max_batch_elems"], g["buckets"])
    assert [[int(i) for i in b] for b in s] == g["batches"]
    assert len(s) == len(g["batches"])


def test_bucketing_sampler_edge_cases():
    class DS:
        lengths = np.array([], dtype=np.int64)

    random.seed(1)
    assert list(ppad.BucketingSampler(DS(), 10, [0, 5, 10])) == []
    DS.lengths = np.array([20, 30])            # all outside the last edge: silently dropped
    assert list(ppad.BucketingSampler(DS(), 10, [0, 5, 10])) == []


def test_schedulers_match_reference(golden):
    g = golden("g9_optim.npz")
    p = [torch.nn.Parameter(torch.zeros(1))]
    for tag, (desc, n) in {"a": ("1cycle_0.0001_0.005", 100), "b": ("1cycle_0.001_0.01", 37)}.items():
        opt = torch.optim.SGD(p, lr=1.0)
        sch = ptrain.make_scheduler(desc, max_steps=n)(opt)
        trace = []
        for s in range(n):
            ptrain.make_step(sch, step=s + 1)
            trace.append(opt.param_groups[0]["lr"])
        np.testing.assert_allclose(trace, g["onecycle_" + tag], rtol=0, atol=1e-18)
    opt = torch.optim.SGD(p, lr=1.0)
    sch = ptrain.make_scheduler("steplr_2_0.5", max_steps=10)(opt)
    assert isinstance(sch, torch.optim.lr_scheduler.StepLR)
    assert set(ptrain.OPTIMIZERS) == {"adam", "momentum"}
    with pytest.raises(ValueError):
        ptrain.make_scheduler("cosine_1_2", 10)


def test_transform_pipeline_and_dataset_protocol():
    class_map = {"b": 1, "a": 0, "c": 2}
    assert putils.get_class_names_from_classmap(class_map) == ["a", "b", "c"]
    tf = ptf.Compose([ptf.SyntheticAudio(), ptf.SampleLongAudio(1), ptf.MapLabels(class_map), ptf.ShuffleAudio(p=0.5),
                      ptf.MixUp(p=1.0), ptf.AudioFeatures("mel_1024_512_64", verbose=False),
                      ptf.DropFields(("audio", "filename", "sr"))])
    clean = ptf.Compose([ptf.SyntheticAudio(), ptf.SampleLongAudio(1), ptf.MapLabels(class_map)])
    files = ["synthetic:%d:%d:16000" % (i, 12000 + 4000 * i) for i in range(4)]
    ds = SoundDataset(files, labels=[["a"], ["b", "c"], ["c"], ["a", "b"]], transform=tf, clean_transform=clean)
    np.random.seed(0)
    random.seed(0)
    item = ds[0]
    assert set(item) == {"signal", "labels", "is_noisy"}
    assert item["signal"].ndim == 2 and item["signal"].shape[1] == 1 and item["signal"].dtype == np.float32
    assert item["labels"].dtype == np.float32 and item["labels"].max() == 1.0
    assert item["signal"].shape[0] <= 16000
    tf.switch_off_augmentations()
    assert all(t.p == 0.0 for t in tf.transforms if isinstance(t, ptf.Augmentation))
    plain = ds[1]
    np.testing.assert_array_equal(plain["labels"], np.array([0, 1, 1], np.float32))
    loader = torch.utils.data.DataLoader(ds, batch_size=4, collate_fn=ppad.make_collate_fn({"signal": 0.0}))
    batch = next(iter(loader))
    assert batch["signal"].shape == (4, 16000, 1) and batch["signal"].dtype == torch.float32
    assert batch["labels"].shape == (4, 3)
    feats = ptf.AudioFeatures("stft_256_128", verbose=False)
    assert feats.n_features == 129 and feats.padding_value == 0.0
    assert ptf.AudioFeatures("mel_2048_1024_128", verbose=False).n_features == 128


def test_shuffle_audio_is_a_permutation_of_chunks():
    random.seed(3)
    x = np.arange(16000, dtype=np.float32)
    y = paudio.shuffle_audio(x.copy(), 0.25, sr=16000)       # 4 chunks
    assert sorted(y.tolist()) == x.tolist() and not np.array_equal(x, y)
    assert paudio.shuffle_audio(x, 2.0, sr=16000) is x       # fewer than two chunks: untouched


def test_l16_plans_cover_the_cfg2_training_layers():
    """The pre-split (L16) kernels have a tiling for the cfg-2 convolutions that carry the step (planning runs on the host):
    forward, input gradient and weight gradient of every 3x3 / 1x1 layer from 100 to 506 channels; the stem (2 input
    channels) and shapes the format cannot serve are declined, not mis-planned."""
    import ctypes

    lib = _lib.load()
    buf = ctypes.create_string_buffer(256)
    layers, hh, ww, cin = [], 128, 431, 2
    for depth in [int(1.5 ** i * 100) for i in range(6)]:          # classifiers.py:524-536, 72-104 at cfg 2
        layers.append((cin, depth, hh, ww, 3))
        hh, ww = hh // 2, ww // 2
        layers += [(depth, depth, hh, ww, 1), (depth, depth, hh, ww, 3)]
        cin = depth
    for (c_in, c_out, h, w, k) in layers:
        d = _lib.ConvDesc(128, c_in, c_out, h, w, k, k, 3)
        big = c_in >= 100 and c_out <= 506
        for dgrad in (0, 1):
            ok = lib.fsc_conv_l16_supported(ctypes.byref(d), dgrad)
            if big:
                assert ok, (c_in, c_out, h, w, k, dgrad)
            if c_in < 32:
                assert not ok
            if ok:
                assert lib.fsc_conv_l16_packed_floats(ctypes.byref(d), dgrad) > 0
                assert lib.fsc_conv_l16_plan_describe(ctypes.byref(d), dgrad, buf, 256) == 0
                assert buf.value.decode().startswith("conv_l16_fwd_kernel<%d,%d," % (k, k))
        # statistics variant of the forward (and of the pooled entry convolution): a worker keeps one channel block
        for pool in (0, 1):
            lay = (ctypes.c_int * 4)()
            if lib.fsc_conv_l16_stats_layout(ctypes.byref(d), pool, lay):
                workers, blocks, co_blk, order = list(lay)
                assert 0 < workers <= 256 and workers % blocks == 0 and co_blk % 16 == 0 and blocks * co_blk >= c_out
                assert order in (0, 1) and (order == 0 or (blocks > 1 and workers % (8 * blocks) == 0))
                assert not pool or (k == 3 and lib.fsc_conv_l16_pool_supported(ctypes.byref(d)))
            elif big and not pool:
                raise AssertionError("no statistics variant for %s" % ((c_in, c_out, h, w, k),))
        if (c_in, c_out, h) in ((100, 150, 64), (150, 225, 32), (225, 337, 16)):
            assert lib.fsc_conv_l16_pool_supported(ctypes.byref(d))       # blocks 1-3: conv + max-pool in one kernel
        okw = lib.fsc_conv_l16_wgrad_supported(ctypes.byref(d))
        if c_in >= 100 and c_out <= 337:
            assert okw, (c_in, c_out, h, w, k)
        if okw:
            assert lib.fsc_conv_l16_wgrad_workspace_bytes(ctypes.byref(d)) > 0
            assert lib.fsc_conv_l16_wgrad_plan_describe(ctypes.byref(d), buf, 256) == 0
    # a descriptor in another arithmetic is not served by the L16 kernels
    assert not lib.fsc_conv_l16_supported(ctypes.byref(_lib.ConvDesc(128, 100, 100, 64, 215, 3, 3, 0)), 0)
    assert lib.fsc_l16_bytes(3, 13, 35) == 3 * 2 * 2 * 35 * 16
