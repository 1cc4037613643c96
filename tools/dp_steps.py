"""Per-step wall time of the cfg-2 training step with and without the data-parallel path in a 1-rank group (development tool)."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from freesound_classification_amd.networks.classifiers import TwoDimensionalCNNClassificationModel  # noqa: E402
from freesound_classification_amd.ops.training import make_step  # noqa: E402

dp = os.environ.get("FSC_FORCE_DP") == "1"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if dp:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group("nccl", device_id=dev)
w = bench.WORKLOADS["cfg2"]
torch.manual_seed(42)
model = TwoDimensionalCNNClassificationModel(bench.make_experiment(w), device="cuda:0")
model.train(); model.global_step = 0
model.make_optimizer(max_steps=40)
signal, labels = bench.synthetic_batch(w, w["batch"], dev, 1234)
if dp and os.environ.get("DP_WARM"):
    for _ in range(int(os.environ["DP_WARM"])):
        for b in model._reducer.buckets:
            dist.all_reduce(b["flat"])
    torch.cuda.synchronize()
    for b in model._reducer.buckets:
        b["flat"].zero_()
if os.environ.get("GC_FREEZE"):
    import gc
    gc.collect(); gc.freeze()
import cProfile, pstats, io
prof = cProfile.Profile()
ts = []
for i in range(16):
    if i == 2:
        prof.enable()
    if i == 12:
        prof.disable()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.global_step += 1; make_step(model.scheduler, step=model.global_step)
    model.training_step(signal, labels)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
print("dp" if dp else "plain", " ".join("%.0f/%.0f" % t for t in ts), "(host ms / total ms per step)")
st = io.StringIO(); pstats.Stats(prof, stream=st).sort_stats("tottime").print_stats(12); print(st.getvalue())
if dp:
    print("in place:", model._reducer.in_place, "of", len(model._reducer.params))
