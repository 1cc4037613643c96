"""Data parallelism on the real model without an 8-GPU node: two replicas (two processes, `gloo` rendezvous, both on
the one visible MI355X) each take half of a seeded global batch.  With cross-replica BatchNorm statistics (SyncBN) the
data-parallel step must equal the single-process step on the concatenated batch -- SURVEY.md section 4's oracle "N
replicas == single process on the concatenated batch" -- within 1e-3: logits, every gradient, BN running statistics,
parameters after the Adam step.  With local BN (the default, what cfg 4 times) each replica must equal a
single-process run on its own shard."""
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

from freesound_classification_amd import functional as F  # noqa: E402
from freesound_classification_amd.networks.classifiers import TwoDimensionalCNNClassificationModel  # noqa: E402
from freesound_classification_amd.networks.losses import lsep_loss  # noqa: E402
import dp_worker  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_two_replicas(outdir, sync_bn, tail=True):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FSC_DP_TEST_TAIL="1" if tail else "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dp_worker.py"), str(outdir), "1" if sync_bn else "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return [dict(np.load(os.path.join(outdir, "rank%d.npz" % k))) for k in (0, 1)]


def _single_process(x, y, seed):
    torch.manual_seed(seed)
    m = TwoDimensionalCNNClassificationModel(dp_worker.make_experiment(False), device="cuda:0")
    m.train()
    m.make_optimizer(max_steps=10)
    assert m._reducer is None
    logits = m(x.cuda())["class_logits"]
    per = lsep_loss(logits, y.cuda(), average=False)
    F.mean(per).backward()
    grads = {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters()}
    for grp in m.optimizer.param_groups:
        grp["lr"] = 1e-3
    m.optimizer.step()
    torch.cuda.synchronize()
    return logits.detach().cpu().numpy(), per.detach().cpu().numpy(), grads, {k: v.cpu().numpy() for k, v in m.state_dict().items()}


def _compare_states(got, want, grads_want, flipped=False):
    for k, v in want.items():
        g = grads_want.get(k)
        noise = g is not None and np.abs(g).max() < 1e-5          # analytically-zero gradients: Adam turns noise into +-lr
        tol = 4e-3 if noise else (2e-3 if "running_mean" in k else 1e-3)
        if flipped and g is not None:                             # (the first Adam step is lr * sign(g): an element whose gradient
            tol = max(tol, 2.5e-3)                                #  changed sign under the PReLU event moves by 2 lr = 2e-3)
        assert np.abs(got["state." + k].astype(np.float64) - v).max() < tol, k


@pytest.mark.timeout(900)
@pytest.mark.parametrize("tail", [False, True], ids=["full_clips", "zero_padded_tail"])
def test_two_replicas_with_syncbn_equal_the_single_process_global_batch(tmp_path, tail):
    """Logits, per-sample losses, BatchNorm buffers and parameters after the step: within 1e-3 in both variants.  Gradients: on
    the batch of full clips EVERY element within 1e-3 (measured 5e-6).  On the batch with a zero-padded tail -- round 6, found
    when the default arithmetic became fp32-exact (tools/dbg_dp_syncbn.py, profiles/r06_dp_syncbn_prelu_flip.txt): the two
    configurations take every max-pool and global-max decision alike (188 416 + 67 584 + 23 040 windows, 960 planes compared)
    and agree to 3e-6 on every backward tensor until block 1's bn3 + residual + PReLU, where ONE activation of 67 584 (clip 0,
    channel 29, position (2, 6)) lies within fp32 rounding of zero and the cross-replica statistics (equal to 1e-7 relative, not
    bit for bit: two fp64 partial sums added in another order) put it on the other side of the PReLU kink: its gradient changes
    by the factor 1 / alpha, which moves the 48 weight gradients of that output channel of the 1x1 convolution in front by 2e-3
    and everything upstream by up to 3.7e-3 of its maximum.  Both are exact evaluations of the same function on either side of
    a measure-zero set -- the single process agrees with the fp64 oracle there, the CPU reference in fp32 has its own such
    events.  So the bound on this variant is the north star's per-tensor statement: rms within 1e-3 of the tensor's scale for
    every tensor (measured <= 3e-4) and >= 99.9 % of ALL gradient elements within 1e-3."""
    r0, r1 = _run_two_replicas(tmp_path, sync_bn=True, tail=tail)
    x, y = dp_worker.global_batch(tail)
    logits, per, grads, state = _single_process(x, y, seed=5)        # rank 0's seed: its parameters were broadcast
    assert np.abs(np.concatenate([r0["logits"], r1["logits"]]) - logits).max() < 1e-3
    assert np.abs(np.concatenate([r0["per"], r1["per"]]) - per).max() < 1e-3
    beyond = total = 0
    worst = ("", 0.0, 0.0)
    for k, g in grads.items():
        assert np.array_equal(r0["grad." + k], r1["grad." + k]), k             # both replicas hold the reduced gradient
        d = np.abs(r0["grad." + k].astype(np.float64) - g) / max(1.0, float(np.abs(g).max()))
        rms = float(np.sqrt((d ** 2).mean()))
        beyond += int((d > 1e-3).sum())
        total += d.size
        if float(d.max()) > worst[1]:
            worst = (k, float(d.max()), rms)
        if tail:
            assert rms < 1e-3, (k, rms)
        else:
            assert float(np.abs(r0["grad." + k] - g).max()) < 1e-3, (k, float(np.abs(r0["grad." + k] - g).max()))
    from test_cfg2_gpu import _report
    _report("two replicas with SyncBN vs the single-process global batch (%s): worst gradient element %.2e of its tensor's scale (%s, "
            "rms %.2e); %d of %d gradient elements beyond 1e-3" % ("zero-padded tail" if tail else "full clips", worst[1], worst[0],
                                                                    worst[2], beyond, total))
    assert beyond <= 1e-3 * total, (beyond, total)
    for r in (r0, r1):
        _compare_states(r, state, grads, flipped=tail)
    for k in state:                                                    # replicas stay in lock-step
        assert np.array_equal(r0["state." + k], r1["state." + k]), k
    n_bn = sum(1 for k in state if k.endswith("running_mean"))
    # one small all-reduce per BN layer and direction -- except the backward of the first block's input BN, which does not
    # run: its parameter gradients come from the stem weight gradient, local sums like every dgamma / dbeta
    assert int(r0["bn_sync_calls"]) == 2 * n_bn - 1
    # (every gradient's slot in a bucket is padded to 16 bytes)
    assert int(r0["bucket_sizes"].sum()) == sum((v.size + 3) // 4 * 4 for v in grads.values())


@pytest.mark.timeout(900)
def test_two_replicas_with_local_bn_equal_independent_shards(tmp_path):
    """Default mode (what cfg 4 times): BatchNorm statistics stay local, so before the gradient exchange every replica
    is exactly the reference at the per-GPU batch; the exchanged gradient is the mean of the two shard gradients."""
    r0, r1 = _run_two_replicas(tmp_path, sync_bn=False)
    x, y = dp_worker.global_batch()
    shard = [_single_process(x[lo:hi], y[lo:hi], seed=5) for lo, hi in ((0, 4), (4, 8))]
    for r, (logits, per, _, _) in zip((r0, r1), shard):
        assert np.abs(r["logits"] - logits).max() < 1e-3
    for k in shard[0][2]:
        mean = (shard[0][2][k] + shard[1][2][k]) / 2
        assert np.abs(r0["grad." + k] - mean).max() < 1e-3, k
        assert np.array_equal(r0["grad." + k], r1["grad." + k]), k
    assert int(r0["bn_sync_calls"]) == 0
