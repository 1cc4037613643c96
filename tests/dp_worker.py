"""Worker of tests/test_dp_gpu.py: one data-parallel replica of the REAL model (launched twice by
torch.distributed.run, `gloo` rendezvous, both replicas on the single visible GPU).  Runs one training step on its
contiguous shard of a seeded global batch with cross-replica BatchNorm and the bucketed gradient all-reduce, and writes
what it saw to <outdir>/rank<r>.npz."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class NS(dict):
    __getattr__ = dict.__getitem__


def make_experiment(sync_bn):
    return NS(config=NS(
        network=NS(num_conv_blocks=3, start_deep_supervision_on=1, conv_base_depth=32, growth_rate=1.5,
                   output_dropout=0.0, aggregation_type="max"),
        data=NS(features="mel_1024_512_64", _input_dim=64, _n_classes=80),
        train=NS(accumulation_steps=1, optimizer="adam", learning_rate=1e-3, weight_decay=0.0,
                 scheduler="1cycle_0.0001_0.005", sync_bn=sync_bn)))


def global_batch(tail=None):
    """Seeded global batch of 8 clips; `tail` (default: FSC_DP_TEST_TAIL, on): clip 3 ends in a zero-padded tail, as collated
    batches do (ops/padding.py:11-31) -- its frames are live data at log(1e-4)."""
    g = torch.Generator().manual_seed(77)
    x = 0.1 * torch.randn(8, 24000, 1, generator=g)
    if (os.environ.get("FSC_DP_TEST_TAIL", "1") == "1") if tail is None else tail:
        x[3, 15000:] = 0.0
    y = torch.zeros(8, 80)
    y[torch.arange(8), torch.randint(0, 80, (8,), generator=g)] = 1.0
    y[2, 5] = 1.0
    return x, y


def main():
    outdir, sync_bn = sys.argv[1], sys.argv[2] == "1"
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from freesound_classification_amd import functional as F
    from freesound_classification_amd import parallel
    from freesound_classification_amd.networks.classifiers import TwoDimensionalCNNClassificationModel
    from freesound_classification_amd.networks.losses import lsep_loss
    if os.environ.get("FSC_DP_DUMP"):                 # (development: tools/dbg_dp_syncbn.py compares the discrete decisions)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import dp_dump
        dp_dump.install()
    torch.manual_seed(5 + rank)                       # replicas start DIFFERENT: the broadcast must fix it
    model = TwoDimensionalCNNClassificationModel(make_experiment(sync_bn), device="cuda:0")
    model.train()
    model.make_optimizer(max_steps=10)
    assert model._reducer is not None and (model._bn_sync is not None) == sync_bn
    x, y = global_batch()
    lo, hi = parallel.shard_range(8)
    xs, ys = x[lo:hi].cuda(), y[lo:hi].cuda()
    logits = model(xs)["class_logits"]
    per = lsep_loss(logits, ys, average=False)
    loss = F.mean(per)
    model._reducer.prepare(sync=True)
    loss.backward()
    model._reducer.finish()
    if os.environ.get("FSC_DP_DUMP"):
        dp_dump.save(os.environ["FSC_DP_DUMP"] + ".rank%d.npz" % rank)
    out = {"logits": logits.detach().cpu().numpy(), "per": per.detach().cpu().numpy(),
           "bn_sync_calls": np.int64(model._bn_sync.calls if sync_bn else 0),
           "bucket_sizes": np.asarray(model._reducer.bucket_sizes(), np.int64)}
    for k, p in model.named_parameters():
        out["grad." + k] = (p.grad / world).detach().cpu().numpy()
    for grp in model.optimizer.param_groups:
        grp["lr"] = 1e-3
    model.optimizer.step()
    torch.cuda.synchronize()
    for k, v in model.state_dict().items():
        out["state." + k] = v.detach().cpu().numpy()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
