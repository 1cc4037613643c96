"""Data-parallel glue on CPU: two `gloo` ranks.  The gradient reducer must make
sum_ranks(grad) / world equal the gradient of the global-mean loss on the concatenated batch,
bucket by bucket, including gradient accumulation (no communication on non-stepping
micro-batches)."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from freesound_classification_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _model():
    torch.manual_seed(7)
    return torch.nn.Sequential(torch.nn.Linear(12, 33), torch.nn.PReLU(33), torch.nn.Linear(33, 5))


def _worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _model()
        if rank == 1:                               # replicas start different: broadcast must fix it
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(1.0)
        parallel.broadcast_module(model)
        reducer = parallel.BucketedGradReducer(list(model.parameters()), bucket_bytes=600)
        assert len(reducer.buckets) >= 2 and sum(reducer.bucket_sizes()) == sum(p.numel() for p in model.parameters())
        g = torch.Generator().manual_seed(100)
        x = torch.randn(8, 12, generator=g)
        y = torch.randn(8, 5, generator=g)
        lo, hi = parallel.shard_range(8)
        # ---- plain step
        reducer.prepare(sync=True)
        ((model(x[lo:hi]) - y[lo:hi]) ** 2).mean().backward()
        reducer.finish()
        grads = [p.grad.clone() / world for p in model.parameters()]
        # ---- accumulation: first micro-batch local only, second one communicates the sum
        model.zero_grad()
        reducer.prepare(sync=False)
        ((model(x[lo:hi]) - y[lo:hi]) ** 2).mean().backward()
        reducer.finish()
        local_only = [p.grad.clone() for p in model.parameters()]
        reducer.prepare(sync=True)
        ((model(x[lo:hi] * 0.5) - y[lo:hi]) ** 2).mean().backward()
        reducer.finish()
        acc = [p.grad.clone() / world for p in model.parameters()]
        torch.save(dict(grads=grads, acc=acc, local_only=local_only,
                        params=[p.detach().clone() for p in model.parameters()]),
                   os.path.join(outdir, "rank%d.pt" % rank))
        reducer.remove()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_allreduce_equals_global_batch():
    world = 2
    port = _free_port()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, port, d), nprocs=world, join=True)
        r0 = torch.load(os.path.join(d, "rank0.pt"))
        r1 = torch.load(os.path.join(d, "rank1.pt"))
    model = _model()
    for a, b, p in zip(r0["params"], r1["params"], model.parameters()):
        assert torch.equal(a, b) and torch.equal(a, p.detach())       # broadcast from rank 0
    g = torch.Generator().manual_seed(100)
    x = torch.randn(8, 12, generator=g)
    y = torch.randn(8, 5, generator=g)
    ((model(x) - y) ** 2).mean().backward()
    for got0, got1, p in zip(r0["grads"], r1["grads"], model.parameters()):
        assert torch.equal(got0, got1)
        np.testing.assert_allclose(got0.numpy(), p.grad.numpy(), atol=1e-6)
    ref_first = [p.grad.clone() for p in model.parameters()]
    ((model(x * 0.5) - y) ** 2).mean().backward()
    for got, p in zip(r0["acc"], model.parameters()):
        np.testing.assert_allclose(got.numpy(), p.grad.numpy(), atol=1e-6)
    # the non-communicating micro-batch left rank-local gradients (they differ between ranks)
    assert any(not torch.allclose(a, b) for a, b in zip(r0["local_only"], r1["local_only"]))
    del ref_first


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 128, 1001):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert parallel.world_size() == 1 and parallel.rank() == 0
