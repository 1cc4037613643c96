"""Development: why is the step with the batch arriving from the host slower than the resident one?  Times the bench.py h2d loop
under a few switches."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from freesound_classification_amd import functional as F
from freesound_classification_amd.networks.classifiers import TwoDimensionalCNNClassificationModel
from freesound_classification_amd.ops.training import make_step

w = bench.WORKLOADS["cfg2"]
device = torch.device("cuda", 0)
torch.manual_seed(42)
model = TwoDimensionalCNNClassificationModel(bench.make_experiment(w), device=str(device))
model.train(); model.global_step = 0
model.make_optimizer(max_steps=200)
signal, labels = bench.synthetic_batch(w, w["batch"], device, 1234)


def plain(k):
    for _ in range(k):
        model.global_step += 1
        make_step(model.scheduler, step=model.global_step)
        model.training_step(signal, labels)


def timeit(fn, k):
    fn(2); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(k); torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / k


pinned = signal.cpu().pin_memory()
print("pinned:", pinned.is_pinned())
dev_buf = [torch.empty_like(signal), torch.empty_like(signal)]
copy_stream = torch.cuda.Stream(device=device)
events = [None, None]


def upload(slot):
    copy_stream.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(copy_stream):
        dev_buf[slot].copy_(pinned, non_blocking=True)
        ev = torch.cuda.Event(); ev.record(copy_stream)
    events[slot] = ev


def h2d(k, do_copy=True, alternate=True):
    if do_copy: upload(0)
    for i in range(k):
        slot = (i & 1) if alternate else 0
        if do_copy:
            torch.cuda.current_stream(device).wait_event(events[slot]); upload(slot ^ 1)
        model.global_step += 1
        make_step(model.scheduler, step=model.global_step)
        model.training_step(dev_buf[slot], labels)


dev_buf[0].copy_(signal); dev_buf[1].copy_(signal)
print("resident, one tensor        %.2f ms" % timeit(plain, 10))
print("resident, alternating bufs  %.2f ms" % timeit(lambda k: h2d(k, False, True), 10))
print("h2d double-buffered         %.2f ms" % timeit(lambda k: h2d(k, True, True), 10))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record(); dev_buf[0].copy_(pinned, non_blocking=True); e1.record(); torch.cuda.synchronize()
print("one H2D copy of %d MB alone: %.2f ms" % (pinned.numel() * 4 >> 20, e0.elapsed_time(e1)))
F.MULTI_PACK = False
print("h2d, per-layer packing      %.2f ms" % timeit(lambda k: h2d(k, True, True), 10))
F.MULTI_PACK = True
F.set_conv_arith(0)
print("native fp32 steps           %.2f ms" % timeit(plain, 4))
F.set_conv_arith(3)
print("after the fp32 steps: resident %.2f ms" % timeit(plain, 10))
print("after the fp32 steps: h2d      %.2f ms" % timeit(lambda k: h2d(k, True, True), 10))
print("reserved %.1f GB allocated %.1f GB" % (torch.cuda.memory_reserved() / 1e9, torch.cuda.memory_allocated() / 1e9))
torch.cuda.empty_cache()
print("after empty_cache: resident %.2f ms" % timeit(plain, 10))
print("after empty_cache: h2d      %.2f ms" % timeit(lambda k: h2d(k, True, True), 10))
print("reserved %.1f GB" % (torch.cuda.memory_reserved() / 1e9))
