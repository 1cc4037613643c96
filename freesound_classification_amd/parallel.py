"""Data-parallel glue for the training step: one process per GPU, `torch.distributed` with the
"nccl" backend (= RCCL over xGMI on ROCm), or "gloo" for CPU tests.

The reference is single-process (train_2d_cnn.py:355-362); data parallelism is the one
exchange step the hot path needs: a sum all-reduce of the 21.5 M fp32
gradients per step.  Gradients are packed into a few flat buckets in reverse registration
order (the head and the deepest block finish their backward first, and own the three largest
tensors), and each bucket's all-reduce is launched from the autograd hook of its last
gradient on a side stream, so communication overlaps the rest of backward.  xGMI is
point-to-point (ring collectives are per-link bound), hence few large buckets rather than
many small ones.  The 1/world factor is folded into the fused optimizer (`grad_scale`).
"""
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(script, argv, n):
    """Start `n` ranks of `script argv...` on this node -- one process per GPU, rendezvous on 127.0.0.1 -- through
    torch.distributed.run and return the launcher's exit code.  What `python bench.py --gpus N` / `train_2d_cnn.py --gpus N` /
    `predict_2d_cnn.py --gpus N` do when no launcher environment (WORLD_SIZE) is present; the children see RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR, MASTER_PORT as under any torchrun launch."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(script)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n)))
    return subprocess.call(cmd, env=env)


def initialized():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if initialized() else 1


def rank():
    return dist.get_rank() if initialized() else 0


def broadcast_module(module, src=0):
    """Make every replica start from rank `src`'s parameters and buffers."""
    if not initialized():
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src)


def _staged(group):
    """gloo has no device collectives in every build: stage device tensors through the host for it (CPU tests and the
    two-replicas-on-one-GPU parity test); RCCL ("nccl") reduces device memory directly."""
    return dist.get_backend(group) == "gloo"


def all_reduce_sum(t, group=None, async_op=False):
    """In-place sum all-reduce of a tensor over the replicas; returns a handle with .wait() when async_op."""
    if t.is_cuda and _staged(group):
        host = t.detach().cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host)

        class _Done:
            def wait(self):
                return True
        return _Done() if async_op else None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


class SyncBN:
    """Cross-replica BatchNorm statistics: callable handed to the BN wrappers, which call it between
    the two phases of fsc_bn_train_stats / fsc_bn_act_bwd* on the per-channel fp64 sums [sum, sum, count, 0].
    One small sum-all-reduce per BatchNorm layer per direction (32 + 32 per step at cfg 2, <= 4 * 1977 doubles
    each: latency-bound, on the compute stream because the very next kernel consumes the result)."""

    def __init__(self, group=None):
        self.group = group
        self.calls = 0

    def __call__(self, sums):
        self.calls += 1
        all_reduce_sum(sums, self.group)


def shard_range(total, world=None, index=None):
    """Contiguous [lo, hi) slice of `total` units owned by rank `index`; sizes differ by <= 1."""
    world = world_size() if world is None else world
    index = rank() if index is None else index
    base, extra = divmod(total, world)
    lo = index * base + min(index, extra)
    return lo, lo + base + (1 if index < extra else 0)


def gather_rows(order, local, width, device=None, dst=0, group=None):
    """Sharded inference (SURVEY 8e "Inference (cfg5)"): every rank holds `local` = (len(order), width) fp32 rows and their
    dataset indices `order`; rank `dst` receives [(indices, rows)] of every rank (the others an empty list).  ONE tensor
    all-gather of the padded (rows | index) blocks -- no pickling; counts differ per rank and may be zero.  Over RCCL ("nccl")
    the blocks are device tensors, over gloo host tensors."""
    import numpy as np
    world = dist.get_world_size(group)
    staged = _staged(group)
    dev = torch.device("cpu") if staged else torch.device(device if device is not None else "cuda")
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    counts[dist.get_rank(group)] = len(order)
    dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    counts = counts.cpu().tolist()
    cap = max(1, max(counts))
    # the index column travels as two fp32 halves of 2^20 each (exact for up to 2^40 rows): one dtype, one collective
    block = torch.zeros(cap, width + 2, dtype=torch.float32)
    if len(order):
        idx = np.asarray(order, dtype=np.int64)
        block[:len(order), :width] = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float32))
        block[:len(order), width] = torch.from_numpy((idx >> 20).astype(np.float32))
        block[:len(order), width + 1] = torch.from_numpy((idx & ((1 << 20) - 1)).astype(np.float32))
    block = block.to(dev)
    blocks = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(blocks, block, group=group)
    if dist.get_rank(group) != dst:
        return []
    out = []
    for r in range(world):
        b = blocks[r][:counts[r]].cpu().numpy()
        idx = (b[:, width].astype(np.int64) << 20) | b[:, width + 1].astype(np.int64)
        out.append((idx, b[:, :width]))
    return out


class BucketedGradReducer:
    """Sum-all-reduce of parameter gradients in flat buckets, overlapped with backward."""

    def __init__(self, params, bucket_bytes=32 << 20, group=None):
        self.group = group
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []          # list of dict(flat, items=[(param, offset, numel)], pending, handle)
        self._where = {}
        order = list(reversed(self.params))
        cur, cur_elems = [], 0
        limit = max(1, bucket_bytes // 4)
        for p in order:
            if cur and cur_elems + self._slot(p) > limit:
                self._close(cur)
                cur, cur_elems = [], 0
            cur.append(p)
            cur_elems += self._slot(p)
        if cur:
            self._close(cur)
        self._sync = False
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._comm_stream = None
        self._by_ptr = {p.data_ptr(): p for p in self.params}
        self.in_place = 0          # gradients that arrived already inside their bucket (grad_view), since prepare()

    @staticmethod
    def _slot(p):
        """Elements a parameter occupies in its bucket: rounded up to 16 bytes, so that every gradient view starts on a 16-byte
        boundary (the optimizer and the reduce kernels take their 16-byte paths on gradients that alias a bucket; the pad stays 0)."""
        return (p.numel() + 3) // 4 * 4

    def _close(self, plist):
        total = sum(self._slot(p) for p in plist)
        flat = torch.zeros(total, device=plist[0].device, dtype=plist[0].dtype)
        items, off = [], 0
        for p in plist:
            items.append((p, off, p.numel()))
            self._where[id(p)] = (len(self.buckets), off)
            off += self._slot(p)
        self.buckets.append(dict(flat=flat, items=items, pending=len(items), handle=None, launched=False, copy_dst=[], copy_src=[]))

    def bucket_sizes(self):
        return [b["flat"].numel() for b in self.buckets]

    def prepare(self, sync=True):
        """Call before backward.  sync=False (gradient accumulation micro-step): no communication."""
        self._sync = sync
        self.in_place = 0
        for b in self.buckets:
            b["pending"] = len(b["items"])
            b["handle"] = None
            b["launched"] = False
            b["copy_dst"], b["copy_src"] = [], []

    def grad_view(self, weight):
        """The slice of its bucket where the gradient of `weight` (a parameter or a detached alias of one) belongs, shaped
        like it -- or None when the gradient cannot be produced in place: no all-reduce this step (accumulation micro-step)
        or a gradient is already being accumulated.  A backward kernel that writes its result there and returns this very
        tensor saves the flat copy (functional.GRAD_OUT; the 86 MB of convolution weights of cfg 2)."""
        if not self._sync:
            return None
        p = self._by_ptr.get(weight.data_ptr())
        if p is None or p.grad is not None or tuple(p.shape) != tuple(weight.shape):
            return None
        bi, off = self._where[id(p)]
        return self.buckets[bi]["flat"][off:off + p.numel()].view(p.shape)

    @staticmethod
    def _flush_copies(b):
        if b["copy_dst"]:
            torch._foreach_copy_(b["copy_dst"], b["copy_src"])
            b["copy_dst"], b["copy_src"] = [], []

    def _launch(self, b):
        """Stream ordering of a bucket (why no further fences are needed):
        * backward kernels and the multi-tensor copy write `flat` on the compute stream; the comm stream waits for the
          compute stream (`wait_stream`) BEFORE the all-reduce is enqueued, so the collective reads finished gradients;
          under RCCL the collective itself runs on the process group's internal stream, which in turn waits for the comm
          stream at enqueue time;
        * finish(): `handle.wait()` makes the compute stream wait for the collective (stream-side, no host block), and the
          explicit `wait_stream(comm)` orders it behind anything else queued on the comm stream -- the optimizer kernels that
          read `flat` (through p.grad) are enqueued after both;
        * the NEXT step's weight-gradient kernels write `flat` through grad_view() on the compute stream, i.e. behind that
          optimizer step in stream order, hence behind the collective;
        * `flat` lives as long as the reducer and is never returned to the caching allocator while a step is in flight, so no
          other tensor can be handed its memory; it is still registered with the comm stream once (`record_stream`, below) so
          that dropping the reducer mid-flight cannot recycle a bucket under a running collective."""
        flat = b["flat"]
        if flat.is_cuda:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=flat.device)
                for other in self.buckets:
                    other["flat"].record_stream(self._comm_stream)
            self._comm_stream.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(self._comm_stream):
                b["handle"] = all_reduce_sum(flat, self.group, async_op=True)
        else:
            b["handle"] = all_reduce_sum(flat, self.group, async_op=True)
        b["launched"] = True

    def _on_grad(self, p):
        if not self._sync:
            return
        bi, off = self._where[id(p)]
        b = self.buckets[bi]
        flat = b["flat"]
        if p.grad.data_ptr() == flat.data_ptr() + off * flat.element_size() and p.grad.is_contiguous():
            self.in_place += 1                   # written there by the kernel that computed it
        else:                                    # (the many small BN / bias gradients: ONE multi-tensor copy per bucket)
            b["copy_dst"].append(flat[off:off + p.numel()])
            b["copy_src"].append(p.grad.reshape(-1))
        b["pending"] -= 1
        if b["pending"] == 0:
            self._flush_copies(b)
            self._launch(b)

    def finish(self):
        """Call after backward: waits for every bucket and re-points p.grad at the reduced data."""
        if not self._sync:
            return
        for b in self.buckets:
            if not b["launched"]:
                # some parameter received no gradient this step: contribute zeros for it
                for p, off, n in b["items"]:
                    if p.grad is None:
                        b["flat"][off:off + n].zero_()
                self._flush_copies(b)
                self._launch(b)
        for b in self.buckets:
            b["handle"].wait()
            if b["flat"].is_cuda:
                torch.cuda.current_stream(b["flat"].device).wait_stream(self._comm_stream)
            for p, off, n in b["items"]:
                p.grad = b["flat"][off:off + n].view_as(p)
        self._sync = False

    def remove(self):
        """Detach from the parameters (the hooks hold the reducer, the reducer holds the parameters: a cycle that only this
        call breaks) and give the gradients back their own storage, so the buckets can be freed."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for b in self.buckets:
            for p, _off, _n in b["items"]:
                if p.grad is not None and p.grad.untyped_storage().data_ptr() == b["flat"].untyped_storage().data_ptr():
                    p.grad = None
        self.buckets = []
        self._where = {}
        self._by_ptr = {}
