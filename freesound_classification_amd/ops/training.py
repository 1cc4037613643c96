"""Counterpart of the reference's ops/training.py: `OPTIMIZERS`, `make_scheduler`, `make_step`,
`OneCycleScheduler` with the same descriptors ("steplr_{step}_{gamma}", "1cycle_{min}_{max}").
The optimizers are fused multi-tensor HIP kernels with torch-compatible state dicts.
"""
import ctypes as C
import gc
import os
import warnings
from functools import partial

import torch
from torch.optim import Optimizer
from torch.optim.lr_scheduler import StepLR

from .. import _lib
from .._lib import OptTensor, call, stream_ptr


def _forget_packed_weights():
    # the kernels below update the weights through raw pointers (no tensor version bump): packed copies kept for inference go
    from .. import functional
    functional.forget_packed_weights()


def _table(entries):
    arr = (OptTensor * len(entries))()
    for i, (p, g, s0, s1, s2) in enumerate(entries):
        arr[i] = OptTensor(p.data_ptr(), g.data_ptr(), s0.data_ptr(),
                           s1.data_ptr() if s1 is not None else None,
                           s2.data_ptr() if s2 is not None else None, p.numel())
    return arr


def _checked_grad(p):
    g = p.grad
    if g.is_sparse:
        raise RuntimeError("sparse gradients are not supported")
    if not p.is_cuda:
        raise _lib.FscError("fused optimizers need device parameters (no CPU fallback)")
    if not g.is_contiguous():
        g = g.contiguous()
        p.grad = g
    return g


class FusedAdam(Optimizer):
    """torch.optim.Adam semantics (L2 weight decay, optional amsgrad) in one kernel per 64
    tensors.  State keys match torch (step, exp_avg, exp_avg_sq, max_exp_avg_sq).
    `grad_scale` multiplies gradients first (1/world_size after a summing all-reduce)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False):
        if not amsgrad:
            raise NotImplementedError("the accelerated path ships Adam with amsgrad=True only "
                                      "(reference ops/training.py:10)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad))
        self.grad_scale = 1.0
        self.factors_dev = None      # set while a step is recorded into a HIP graph (CapturedTrainingStep): lr / step from device memory

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        _forget_packed_weights()
        for group in self.param_groups:
            entries = []
            step = None
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = _checked_grad(p)
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["max_exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] = int(st["step"]) + 1
                if step is None:
                    step = st["step"]
                if st["step"] != step:       # tensors that joined late get their own launch
                    self._launch(group, [(p, g, st["exp_avg"], st["exp_avg_sq"], st["max_exp_avg_sq"])], st["step"])
                    continue
                entries.append((p, g, st["exp_avg"], st["exp_avg_sq"], st["max_exp_avg_sq"]))
            if entries:
                self._launch(group, entries, step)
        return loss

    def _launch(self, group, entries, step):
        b1, b2 = group["betas"]
        tbl = _table(entries)
        from .. import functional as F
        # reads p, g, m, v, vmax and writes p, m, v, vmax: nine fp32 streams (776 MB per step at cfg 2)
        with F._stage("optimizer", 9 * 4 * sum(e[0].numel() for e in entries)):
            if self.factors_dev is not None:
                call("fsc_adam_amsgrad_step_dev", tbl, len(entries), self.factors_dev.data_ptr(), b1, b2, group["eps"],
                     group["weight_decay"], float(self.grad_scale), stream_ptr())
            else:
                call("fsc_adam_amsgrad_step", tbl, len(entries), float(group["lr"]), b1, b2, group["eps"],
                     group["weight_decay"], int(step), float(self.grad_scale), stream_ptr())


class FusedSGD(Optimizer):
    """torch.optim.SGD(momentum, nesterov=True, dampening=0) semantics with L2 weight decay."""

    def __init__(self, params, lr, momentum=0.9, weight_decay=0.0, nesterov=True):
        if not nesterov or momentum <= 0:
            raise NotImplementedError("the accelerated path ships SGD with Nesterov momentum only "
                                      "(reference ops/training.py:11)")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=nesterov,
                                      dampening=0))
        self.grad_scale = 1.0

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        _forget_packed_weights()
        for group in self.param_groups:
            fresh, warm = [], []
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = _checked_grad(p)
                st = self.state[p]
                if "momentum_buffer" not in st or st["momentum_buffer"] is None:
                    st["momentum_buffer"] = torch.empty_like(p, memory_format=torch.contiguous_format)
                    fresh.append((p, g, st["momentum_buffer"], None, None))
                else:
                    warm.append((p, g, st["momentum_buffer"], None, None))
            for entries, first in ((fresh, 1), (warm, 0)):
                if entries:
                    call("fsc_sgd_nesterov_step", _table(entries), len(entries), float(group["lr"]),
                         group["momentum"], group["weight_decay"], first, float(self.grad_scale), stream_ptr())
        return loss


OPTIMIZERS = {
    "adam": partial(FusedAdam, amsgrad=True),
    "momentum": partial(FusedSGD, momentum=0.9, nesterov=True),
}


class OneCycleScheduler:
    """Linear warm-up from min_lr to max_lr over the first 30 % of `max_steps`, then linear
    decay to min_lr / 1000 (reference ops/training.py:208-234).  `step()` sets the lr for the
    upcoming optimizer step."""

    def __init__(self, optimizer, min_lr, max_lr, max_steps):
        self.optimizer = optimizer
        self.min_lr = min_lr
        self.max_lr = max_lr
        self.max_steps = max_steps
        self.epoch = -1

    def lr_at(self, index):
        knee = int(round(self.max_steps * 0.3))
        if index < knee:
            return self.min_lr + (index / knee) * (self.max_lr - self.min_lr)
        frac = (index - knee) / (self.max_steps - knee)
        return self.max_lr + frac * (self.min_lr / 1e3 - self.max_lr)

    def step(self):
        self.epoch += 1
        lr = self.lr_at(self.epoch)
        for group in self.optimizer.param_groups:
            group["lr"] = lr


def make_scheduler(params, max_steps):
    """Descriptor -> callable(optimizer) building the scheduler (reference ops/training.py:15-34)."""
    name, *args = params.split("_")
    if name == "steplr":
        step_size, gamma = int(args[0]), float(args[1])
        return partial(StepLR, step_size=step_size, gamma=gamma)
    if name == "1cycle":
        min_lr, max_lr = float(args[0]), float(args[1])
        return partial(OneCycleScheduler, min_lr=min_lr, max_lr=max_lr, max_steps=max_steps)
    raise ValueError("unknown scheduler descriptor %r" % params)


def make_step(scheduler, epoch=None, step=None, val_score=None):
    """StepLR advances per epoch, 1cycle per training step (reference ops/training.py:37-43)."""
    if isinstance(scheduler, StepLR) and epoch is not None:
        scheduler.step(epoch)
    elif isinstance(scheduler, OneCycleScheduler) and step is not None:
        scheduler.step()


class CapturedTrainingStep:
    """One training step of a model -- forward, loss, backward, optimizer -- recorded ONCE as a HIP graph and replayed per batch.

    The same entry points with the same arguments in the same order as the eager step, so a replayed step equals the eager one
    (40 replays of one recording stay within 1.5e-5 of the eager logits over four steps, tools/dbg_graph.py); what it removes is
    the host: ~510 entry-point calls of Python per cfg-3 step.  On a host that issues the step faster than the GPU runs it
    (this one: 9.2 ms against 11.5 ms) that buys nothing -- bench.py keeps eager steps and offers `--graph`.

    What makes it legal: every entry point of libfsc_hip.so is capture-safe (include/fsc_hip.h: no allocation, no
    synchronisation, no host read-back); the step-dependent scalars of Adam travel through two device floats
    (fsc_adam_amsgrad_step_dev) refreshed in front of each replay; BatchNorm counters are bumped by a kernel.  What it needs:
    fixed batch shape, Adam, accumulation_steps == 1, no dropout (its counter-based stream takes the offset by value), single
    GPU (no reducer), at least one eager step before (lazy state exists).  `model.training_step` semantics otherwise: returns
    (class_logits, per-sample loss, loss) -- static tensors, valid until the next call.

    Two rules for the caller, both enforced:
    * ONE NON-DEFAULT STREAM for everything -- the eager steps before, the recording, the replays: construct and call inside the
      same `with torch.cuda.stream(s):`.  With the eager steps on torch's default stream (the legacy null stream) and the graph
      on another, replays were wrong in ways that depended on where the host synchronised (NaN logits within three replays with
      a device synchronise after every step; a stalled loss, or a NaN in one run of six, with bench.py's pattern; correct with a
      read-back after every step) -- tools/dbg_graph5.py reproduces it; nothing in the library or in torch's strictest capture
      mode flags anything, and the same loop on one non-default stream is correct under every pattern tried.
    * NO tensor of an earlier step's autograd graph may be alive at the recording (drop the eager steps' logits / losses first):
      it keeps the parameters' AccumulateGrad nodes alive, which then run outside the recording.  The constructor drops the
      library's own references (the BatchNorm statistics stash), collects garbage, and refuses the recording if torch reports
      such a node.
    """

    def __init__(self, model, signal, labels):
        opt = model.optimizer
        if not isinstance(opt, FusedAdam):
            raise _lib.FscError("CapturedTrainingStep needs the fused Adam optimizer")
        if model._reducer is not None or model.config.train.accumulation_steps != 1:
            raise _lib.FscError("CapturedTrainingStep: single GPU, accumulation_steps == 1")
        if float(model.config.network.output_dropout) > 0.0:
            raise _lib.FscError("CapturedTrainingStep: dropout draws from a host-side counter; capture needs output_dropout == 0")
        if len(opt.param_groups) != 1 or not all(opt.state[p] for p in opt.param_groups[0]["params"] if p.requires_grad):
            raise _lib.FscError("CapturedTrainingStep: run one eager training step first (optimizer state must exist)")
        stream = torch.cuda.current_stream(signal.device)
        if stream == torch.cuda.default_stream(signal.device):
            raise _lib.FscError("CapturedTrainingStep: run the training loop -- eager steps, this constructor, the replays -- inside "
                                "ONE `with torch.cuda.stream(s):` (not on the default stream; see the class docstring)")
        self.stream = stream
        self.model, self.opt = model, opt
        self.group = opt.param_groups[0]
        self.step_count = int(next(iter(opt.state.values()))["step"])
        self.signal = signal.clone()
        self.labels = labels.clone()
        self.factors = torch.zeros(2, device=signal.device, dtype=torch.float32)
        self._buf = (C.c_float * 2)()
        self.graph = torch.cuda.CUDAGraph()
        opt.zero_grad()
        from .. import functional
        functional._PRESTATS.clear()             # (holds the last block output of the previous step, i.e. its autograd graph)
        gc.collect()
        opt.factors_dev = self.factors
        try:
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                with torch.cuda.graph(self.graph, stream=stream, capture_error_mode=os.environ.get("FSC_CAPTURE_MODE", "relaxed")):
                    self.outputs = model.training_step(self.signal, self.labels)
        finally:
            opt.factors_dev = None
        # the recording ran the step's Python (not its kernels): undo the one host-side effect, the step counters
        for st in opt.state.values():
            st["step"] = self.step_count
        stale = [w for w in caught if "AccumulateGrad" in str(w.message)]
        for w in caught:
            if w not in stale:
                warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
        if stale:
            self.graph = None
            raise _lib.FscError("CapturedTrainingStep: a tensor of an earlier step's autograd graph is still alive (its AccumulateGrad "
                                "nodes would run outside the recording); drop the previous steps' outputs before recording")

    def __call__(self, signal, labels):
        if torch.cuda.current_stream(self.signal.device) != self.stream:
            raise _lib.FscError("CapturedTrainingStep: replay on the stream the step was recorded on")
        if tuple(signal.shape) != tuple(self.signal.shape) or tuple(labels.shape) != tuple(self.labels.shape):
            raise _lib.FscError("CapturedTrainingStep: recorded for a batch of shape %s / %s, got %s / %s (copy_ would broadcast)"
                                % (tuple(self.signal.shape), tuple(self.labels.shape), tuple(signal.shape), tuple(labels.shape)))
        self.signal.copy_(signal, non_blocking=True)
        self.labels.copy_(labels, non_blocking=True)
        self.step_count += 1
        # the replay rewrites parameters and BatchNorm buffers through raw pointers (no torch version bump): the caches of an
        # inference forward between replays (packed fragments, folded eval-mode BatchNorm) must not survive it
        from .. import functional
        functional.forget_packed_weights()
        for st in self.opt.state.values():                # (a checkpoint saved mid-run resumes with the right bias correction)
            st["step"] = self.step_count
        b1, b2 = self.group["betas"]
        _lib.load().fsc_adam_step_factors(float(self.group["lr"]), b1, b2, self.step_count, self._buf)
        # (by-value kernel arguments: no host buffer that a later call could overwrite before an earlier copy has run)
        call("fsc_fill", self.factors.data_ptr(), float(self._buf[0]), 1, stream_ptr())
        call("fsc_fill", self.factors.data_ptr() + 4, float(self._buf[1]), 1, stream_ptr())
        self.graph.replay()
        return self.outputs

    def sync_state(self):
        """Write the step count back into the optimizer's state (state_dict compatibility) -- call before saving."""
        for st in self.opt.state.values():
            st["step"] = self.step_count
