"""The inference split over N ranks (predict_2d_cnn.predict_folds; reference predict_2d_cnn.py:72-125, SURVEY 8e
"Inference (cfg5)") without an N-GPU node: N processes with a `gloo` rendezvous, all on the one visible MI355X, each takes
the length-grouped batches rank::world, averages its fold models locally and one tensor gather brings the rows to rank 0.
Rank 0's array must equal the single-process array BIT FOR BIT (same batch composition -> same padded batches -> same
kernels), dataset order preserved -- with uneven batch counts per rank and with a rank that receives no batch at all."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import predict_worker  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_ranks(outdir, world, n_clips, max_batch_seconds, bucket_seconds=0.5, arith=None, fold=True):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "predict_worker.py"), str(outdir),
           str(n_clips), str(max_batch_seconds), str(bucket_seconds), arith or "default", "1" if fold else "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return np.load(os.path.join(outdir, "world%d.npy" % world)), int(np.load(os.path.join(outdir, "nbatches%d.npy" % world))[0])


@pytest.fixture(autouse=True)
def _restore():
    from freesound_classification_amd import functional as F
    mode0, fold0 = F.get_conv_arith(), F.EVAL_ACT_FOLD
    yield
    F.set_conv_arith(mode0)
    F.EVAL_ACT_FOLD = fold0
    F.forget_packed_weights()


# What "the same" means here.  The single process is not bit-reproducible from run to run to begin with: the head's Linear layers
# (and the last blocks' small convolutions) split K over workgroups and add their slices with atomics, so the summation order --
# and the last bit or two of a logit -- changes between two runs of the SAME call (measured below as `noise`, <= 1e-6).  On top of
# that, the FOLDED route of the scaled arithmetics declares each folded operand's maximum from a calibration -- twice the bound of
# the first batch THIS PROCESS saw -- so elements more than 2^-16 below that maximum round differently after a different history
# (a rank that starts at batch 1 instead of batch 0).  Asserted: rank 0 of the N-rank run returns every row of the dataset, in
# dataset order, within max(4 x noise, 2e-6) of the single-process array -- i.e. indistinguishable from a second single-process run.
@pytest.mark.timeout(900)
@pytest.mark.parametrize("arith,fold", [("bf16x9", True), (None, False), (None, True)], ids=["bf16x9_folded", "default_two_pass", "default_folded"])
def test_sharded_inference_returns_rank_0_the_single_process_probabilities(tmp_path, arith, fold):
    from test_cfg2_gpu import _report
    n_clips, max_batch_seconds = 37, 12.0
    single, batches = predict_worker.run(n_clips, max_batch_seconds, arith=arith, fold=fold)       # this process, world = 1
    assert single.shape == (n_clips, 80) and np.isfinite(single).all() and (single > 0).all() and (single < 1).all()
    assert sorted(i for b in batches for i in b) == list(range(n_clips))         # every clip exactly once
    assert len(batches) >= 5 and len(batches) % 2 == 1                           # uneven split over two ranks
    again, _ = predict_worker.run(n_clips, max_batch_seconds, arith=arith, fold=fold)
    noise = float(np.abs(single - again).max())                                  # run-to-run noise of the single process itself
    assert noise <= 1e-6, noise
    two, nb2 = _run_ranks(tmp_path, 2, n_clips, max_batch_seconds, arith=arith, fold=fold)
    assert nb2 == len(batches)
    diff = float(np.abs(two - single).max())
    _report("sharded inference (2 ranks on one GPU, %s, fold %s): rank 0 vs single process %.2e; single process run-to-run %.2e; "
            "%d batches for 2 ranks" % (arith or "default f16x6", fold, diff, noise, len(batches)))
    assert diff <= max(4.0 * noise, 2e-6), (diff, noise)
    # rows are in DATASET order, not batch order: clip i of the dataset is row i whoever computed it (a permuted row would be
    # off by ~1e-1: the clips have different lengths and contents)
    lens = predict_worker.make_dataset(n_clips)[0].lengths
    assert len(lens) == n_clips and not np.array_equal(np.argsort(lens, kind="stable"), np.arange(n_clips))
    assert float(np.abs(single[1:] - single[:-1]).max()) > 1e-3
    # the one-batch-at-a-time entry point gives the same rows as the pipelined generator predict_folds drives
    import predict_2d_cnn as drv
    from freesound_classification_amd import functional as F
    from freesound_classification_amd.ops.padding import make_collate_fn
    F.set_conv_arith(arith)
    F.EVAL_ACT_FOLD = fold
    F.forget_packed_weights()
    dataset, features = predict_worker.make_dataset(n_clips)
    collate = make_collate_fn({"signal": features.padding_value})
    models = predict_worker.make_models()
    rows = np.zeros_like(single)
    for b in batches:
        x = collate([dataset[i] for i in b])["signal"].to("cuda:0")
        rows[b] = drv.ensemble_batch_checked(models, x).numpy()
    assert float(np.abs(rows - single).max()) <= max(4.0 * noise, 2e-6)


@pytest.mark.timeout(900)
def test_sharded_inference_with_a_rank_that_has_no_batch(tmp_path):
    n_clips, max_batch_seconds, bucket = 9, 60.0, 60.0                           # one bucket, two batches at most; three ranks
    single, batches = predict_worker.run(n_clips, max_batch_seconds, bucket, arith="bf16x9")
    assert 1 <= len(batches) <= 2
    three, nb3 = _run_ranks(tmp_path, 3, n_clips, max_batch_seconds, bucket, arith="bf16x9")
    assert nb3 == len(batches)
    assert three.shape == single.shape and float(np.abs(three - single).max()) <= 2e-6
