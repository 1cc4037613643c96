"""Training-step throughput of the accelerated audio-tagging path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = STFT -> mel -> log -> freq channel -> 2-d CNN forward -> per-sample LSEP -> mean ->
backward -> (gradient all-reduce over RCCL when N > 1) -> Adam-amsgrad, on one synthetic batch
already resident in HBM.  Workload = BASELINE.json configs[1]: batch 128 x 10 s @ 44.1 kHz,
mel_2048_1024_128, 6 blocks base 100 growth 1.5, deep supervision from block 1, dropout 0.7
(reference README.md:200-214), fp32.  N > 1: one process per GPU (torch.distributed.run), the
same per-GPU batch on every rank (weak scaling), value = clips of all ranks / max-over-ranks time.

Arithmetic of `value` (cfg 2): fp32 tensors and accumulators, every convolution product formed to 2^-32 -- each operand is split by
the kernel that produces it into three scaled fp16 limbs (24-bit operands, exact for every element within 2^-16 of its tensor's
maximum) and six limb products run on v_mfma_f32_16x16x32_f16 with fp32 accumulation ("f16x6", fsc_conv_desc.arith = 10 -- the
LIBRARY DEFAULT since round 6: `value` is measured with no arithmetic selected anywhere): for an
fp32 accumulator the arithmetic of the reference's nn.Conv2d on fp32 tensors (per-layer errors against fp64 0.1 - 1.4x those of
PyTorch's own fp32 convolution, tests/test_l3_gpu.py).  Timed beside it, outside `value`: `exact_mode` (bf16x9: three exact
bf16 limbs, all nine products -- the product of the two fp32 operands is EXACT whatever their range), `fast_mode` (the opt-in
f16x3: two scaled fp16 limbs, three products, 22-bit products) and `alt_f32` (the native fp32-MFMA kernels).

Rank 0 prints one JSON line carrying `roofline` (dominant kernel = the conv kernel with the largest
total time, FLOPs over HIP-event time measured inside the timed region), at N = 1 `fast_mode` / `alt_f32` (the same
workload re-timed in the other arithmetics, outside the timed region of `value`) and
`cpu_baseline` (the CPU oracle = pure-PyTorch restatement of the reference path, timed on this
box's host cores on a bounded sample: batch 8 of the same model and clip length, 1 + 3 steps).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_F32_MFMA_TFLOPS = 157.3      # v_mfma_f32_16x16x4_f32 (native fp32 kernels)
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense fp16 = bf16 MFMA (the split kernels execute 3 (fp16 limbs) or 6 / 9 (bf16 limbs) MFMA flops per fp32 flop)
PEAK_HBM_GBPS = 8000.0            # HBM3E spec (6290 GB/s measured by a float4 copy)


class NS(dict):
    __getattr__ = dict.__getitem__


WORKLOADS = {
    # BASELINE.json configs[1]
    "cfg2": dict(features="mel_2048_1024_128", blocks=6, base=100, growth=1.5, start=1, dropout=0.7,
                 batch=128, samples=441000, sr=44100, n_mel=128),    # (arithmetic: the library default, f16x6)
    # BASELINE.json configs[2]: 1-d raw-STFT path (win 256), 10-block hierarchical CNN, LSEP + MixUp.
    # hop 128 / base 64 / growth 1.25 are SURVEY section 8d's assumptions.  bf16: conv operands rounded to one bf16
    # value, v_mfma_f32_16x16x32_bf16 with fp32 accumulation; weights / BN statistics / optimizer state fp32 masters
    # (`--arith f32` or `f16x3` re-times the same workload in fp32).
    "cfg3": dict(features="stft_256_128", blocks=10, base=64, growth=1.25, start=1, dropout=0.0,
                 batch=128, samples=441000, sr=44100, n_mel=129, dims=1, mixup=0.5, arith="bf16",
                 # the whole one-cycle schedule is squeezed into the ~26 steps of a bench run: at a 0.005 peak the 10-block model on
                 # white noise blows its logits up -- the largest score gap of a run reaches 150 ... 330 in EVERY trial, and LSEP
                 # (reference networks/losses.py:47-58, un-stabilised) overflows fp32 once a negative class leads a positive one by
                 # 88: loss = inf in 2 / 40 trials with the separate BatchNorm finalisation launches and 4 / 40 with the ticketed
                 # ones (the same within counting noise; 0 / 20 and gaps <= 38 at this 0.001 peak): profiles/r04h_nan_steps.txt,
                 # tools/nan_steps.py.  An overflow of the reference's own loss, not a race; the arithmetic per step does not
                 # depend on the rate
                 scheduler="1cycle_0.0001_0.001"),
    # BASELINE.json configs[0] shape (used for quick checks: --workload cfg1)
    "cfg1": dict(features="mel_1024_512_64", blocks=3, base=32, growth=2, start=1, dropout=0.0,
                 batch=64, samples=32000, sr=16000, n_mel=64),
    # BASELINE.json configs[4]: 5-fold ensemble inference on variable-length, length-grouped batches (SURVEY 8d:
    # 4096 clips, lengths U(0.3 s, 30 s) @ 44.1 kHz seed 7, bucket edges every 2 s, <= 128 x 10 s of samples per
    # batch; the cfg-2 network).  A "step" is one length-grouped batch through all five resident fold models.
    "cfg5": dict(features="mel_2048_1024_128", blocks=6, base=100, growth=1.5, start=1, dropout=0.7,
                 batch=128, samples=441000, sr=44100, n_mel=128, inference=dict(folds=5, clips=4096, seed=7,
                                                                                 min_s=0.3, max_s=30.0, bucket_s=2.0)),
}


def make_experiment(w):
    return NS(config=NS(
        network=NS(num_conv_blocks=w["blocks"], start_deep_supervision_on=w["start"],
                   conv_base_depth=w["base"], growth_rate=w["growth"], output_dropout=w["dropout"],
                   aggregation_type="max"),
        data=NS(features=w["features"], _input_dim=w["n_mel"], _n_classes=80),
        train=NS(accumulation_steps=1, optimizer="adam", learning_rate=3e-3, weight_decay=0.0,
                 scheduler=w.get("scheduler", "1cycle_0.0001_0.005"), switch_off_augmentations_on=10 ** 9, _save_every=10 ** 9)))


def synthetic_batch(w, batch, device, seed):
    gen = torch.Generator(device=device).manual_seed(seed)
    signal = 0.1 * torch.randn(batch, w["samples"], 1, device=device, generator=gen)
    labels = (torch.rand(batch, 80, device=device, generator=gen) < 0.02).float()
    pos = torch.randint(0, 80, (batch,), device=device, generator=gen)
    labels[torch.arange(batch, device=device), pos] = 1.0
    return signal, labels


def physical_cores():
    """Physical cores of this host (sockets x cores per socket from /proc/cpuinfo), else the logical count."""
    try:
        ids = set()
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip() and phys is not None and core is not None:
                    ids.add((phys, core))
        if ids:
            return len(ids)
    except OSError:
        pass
    return os.cpu_count() or 1


def cpu_baseline(w, steps=3, batch=8):
    """The oracle (CPU restatement of the reference path) on this box's host cores, as BASELINE.md section 4 plans
    it: the same model and clip length at batch 8, 1 warm-up + 3 timed steps, median.  Threads = the host's PHYSICAL
    cores, stated in the line (one intra-op thread per hardware thread oversubscribes these shapes badly: 0.07
    clips/s at 256 threads on the round-1 box)."""
    from oracle import ref_torch as oref
    cores = physical_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    if w.get("dims") == 1:
        model = oref.TagCNN1d(w["features"], w["blocks"], w["base"], w["growth"], w["start"], 80,
                              output_dropout=w["dropout"], input_dim=w["n_mel"])
    else:
        model = oref.TagCNN2d(w["features"], w["blocks"], w["base"], w["growth"], w["start"], 80,
                              output_dropout=w["dropout"])
    opt = oref.make_adam(model, 3e-3)
    signal, labels = synthetic_batch(w, batch, torch.device("cpu"), 99)
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        oref.train_step(model, opt, signal, labels)
        times.append(time.perf_counter() - t0)
    times = sorted(times[1:])
    med = times[len(times) // 2]
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return dict(value=batch / med, unit="clips/s", cores=cores, kind="port",
                sample="oracle train step, same model and clip length, batch %d, 1 warm-up + %d timed steps "
                       "(median), %s" % (batch, steps, cpu_model))


def price_kernel(name):
    """(peak TFLOP/s, executed 16-bit MFMA flops per algorithmic fp32 flop, arithmetic) of a timed conv kernel, by its plan name.
    Split kernels (`..._x3_kernel<..., NPROD>`, `conv_l16_*`): every algorithmic fp32 flop costs NPROD 16-bit MFMA flops, so the
    achieved rate counts EXECUTED 16-bit flops and is priced against the dense fp16 / bf16 peak."""
    if "conv_l3_" in name:             # pre-split operands, three limbs: <..., NPROD[,f16][,pool]> limb products
        core = name.rstrip(">").replace(",pool", "")
        f16 = core.endswith(",f16")
        per = int(core.replace(",f16", "").split(",")[-1])
        if f16:
            return PEAK_BF16_MFMA_TFLOPS, per, ("fp32 via three SCALED fp16 limbs per operand, written once by the operand's producer (L16 tensors, 6 B per "
                                                "element; exact for elements down to 2^-16 of the tensor maximum), %d fp16 MFMA products per fp32 product "
                                                "(dropped limb pairs <= 2^-32 of the product), fp32 accumulate" % per)
        return PEAK_BF16_MFMA_TFLOPS, per, ("fp32 with exact products: every operand split exactly into three bf16 limbs by its producer "
                                            "(L16 tensors, 6 B per element), %d bf16 MFMA products per fp32 product%s, fp32 accumulate"
                                            % (per, " (all limb pairs: the product of the two fp32 operands is exact)" if per == 9 else ""))
    if "conv_l16_" in name:            # pre-split (L16) operands: the 2-limb fp16 arithmetic, 3 products
        return PEAK_BF16_MFMA_TFLOPS, 3, ("fp32 via 2-limb fp16 split with per-tensor power-of-two scaling (operands pre-split by their "
                                          "producers: L16 tensors), 3 fp16 MFMA products per fp32 product, fp32 accumulate")
    if "_x3_kernel" in name:
        per = int(name.rstrip(">").split(",")[-1])
        arith = "fp32 via exact 3-limb bf16 split, %d bf16 MFMA products per fp32 product, fp32 accumulate" % per
        if per == 1:
            arith = "bf16 operands (one rounding each), bf16 MFMA, fp32 accumulate"
        if per == 3:                   # (the dense fp16 and bf16 MFMA peaks are equal)
            arith = "fp32 via 2-limb fp16 split with per-tensor power-of-two scaling, 3 fp16 MFMA products per fp32 product, fp32 accumulate"
        return PEAK_BF16_MFMA_TFLOPS, per, arith
    return PEAK_F32_MFMA_TFLOPS, 1, "native fp32 MFMA"


ARITH_LABEL = {0: "f32", 1: "bf16", 3: "f16x3", 6: "bf16x6", 9: "bf16x9", 10: "f16x6"}
# significand bits a conv product keeps: 24 = the product of the fp32 operands before the accumulator's rounding (native fp32 MFMA; all
# nine bf16 limb products: exact; f16x6: 24-bit operands, the dropped limb pairs are <= 2^-32 of the product -- 2^-8 of what the fp32
# accumulator rounds away -- for operands down to 2^-16 of their tensor's maximum, include/fsc_hip.h)
ARITH_BITS = {0: 24, 1: 8, 3: 22, 6: 23, 9: 24, 10: 24}


def rocprof_name_to_timer_name(n):
    """Kernel name as rocprofv3 prints it -> the name the KernelTimer uses (fsc_conv_*_plan_describe), or None."""
    import re
    m = re.search(r"(conv_\w+)<([^>]*)>", n)
    if not m:
        return None
    base, a = m.group(1), [x.strip() for x in m.group(2).split(",")]
    if base == "conv_l3_fwd_kernel":                    # <KH, KW, CT, PTW, NPROD, F16, POOL, STATS>
        return "conv_l3_fwd_kernel<%s>" % ",".join(a[:5] + (["f16"] if a[5:6] == ["true"] else []) + (["pool"] if a[6:7] == ["true"] else []))
    if base == "conv_l16_wgrad_kernel":                 # <KH, KW, MT, CT[, NL, NPROD, F16]>
        if len(a) >= 6 and a[4] == "3":
            return "conv_l3_wgrad_kernel<%s>" % ",".join(a[:4] + [a[5]] + (["f16"] if a[6:7] == ["true"] else []))
        return "conv_l16_wgrad_kernel<%s>" % ",".join(a[:4])
    if base == "conv_l16_fwd_kernel":                   # <KH, KW, COT, PT, POOL, STATS>
        return "conv_l16_fwd_kernel<%s>" % ",".join(a[:4] + (["pool"] if a[4:5] == ["true"] else []))
    if base == "conv_fwd_kernel":
        return "conv_fwd_kernel<%s>" % ",".join(a[:4])
    if base == "conv_wgrad_kernel":
        return "conv_wgrad_kernel<%s>" % ",".join(a[:3] + (["packed"] if a[3:4] == ["true"] else []))
    return "%s<%s>" % (base, ",".join(a))


def measure_traffic(kernel, extra_args, timeout=150):
    """HBM bytes per launch of `kernel` (a KernelTimer name) in this very workload, from rocprofv3 PMC counters: two separate passes
    (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) with --kernel-trace only over a 2-step run of this script in a child
    process, averaged over all launches of the kernel -- as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE counts
    64 B per 128-B request of wide streaming reads on gfx950 (x 2; calibrated on fsc_axpy: 1 048 579 KiB reported for a 2 GiB read),
    WRITE_SIZE x 1, both in KiB.  Returns (bytes or None, note)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    total, launches = 0.0, 0
    tmp = tempfile.mkdtemp(prefix="fsc_traffic_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for counter, mult in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "p", "--", sys.executable,
                   os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timer", "--no-alt",
                   "--no-other"] + list(extra_args)
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, cwd="/tmp", env=env)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s failed (rc %d)" % (counter, r.returncode)
            s, c = 0.0, 0
            for f in files:
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") == counter and rocprof_name_to_timer_name(row["Kernel_Name"]) == kernel:
                            s += float(row["Counter_Value"]) * 1024.0 * mult
                            c += 1
            if c == 0:
                return None, "kernel %s not in the %s pass" % (kernel, counter)
            total += s / c
            launches = c
    except Exception as e:                                   # a failed side measurement must not take the line down
        return None, repr(e)[:200]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return total, ("rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two passes) over `bench.py --steps 2 --warmup 1` of this "
                   "workload in a child process, mean of %d launches; FETCH_SIZE x 2 (gfx950 counts 64 B per 128-B request), WRITE_SIZE x 1"
                   % launches)


def roofline_of(summ, steps):
    """`roofline` of a KernelTimer summary: the conv kernel with the largest total time, its algorithmic FLOPs over its HIP-event
    time x the MFMA flops it executes per algorithmic flop, against the dense peak of the pipe it runs on."""
    fam = {}
    for name, r in summ.items():
        f = fam.setdefault(name.split("<")[0], dict(flops=0.0, ms=0.0))
        f["flops"] += r["flops"]
        f["ms"] += r["ms"]
    dom_name, dom = max(summ.items(), key=lambda kv: kv[1]["ms"])
    achieved = dom["flops"] / dom["ms"] / 1e9          # TFLOP/s
    traffic = None                                     # (measured by rocprofv3 PMC passes of this very run: measure_traffic)
    peak, executed_per_flop, arith = price_kernel(dom_name)
    return {
        "kernel": dom_name, "bound": "mfma", "achieved": achieved * executed_per_flop, "peak": peak,
        "unit": "TFLOP/s", "frac": achieved * executed_per_flop / peak, "traffic": traffic,
        "arithmetic": arith, "algorithmic_fp32_tflops": achieved,
        "launches_per_step": dom["launches"] / steps,
        "avg_launch_ms": dom["ms"] / dom["launches"],
        "algorithmic_gflop_per_launch": dom["flops"] / dom["launches"] / 1e9,
        "conv_ms_per_step": {k: v["ms"] / steps for k, v in fam.items()},
        "conv_tflops": {k: v["flops"] / v["ms"] / 1e9 for k, v in fam.items()},
        "conv_ms_total_per_step": sum(v["ms"] for v in fam.values()) / steps,
    }


def cpu_baseline_inference(w, batch=8, runs=2):
    """cfg 5 on the host cores: the oracle's eval-mode forward of one batch of `batch` 10 s clips through five fold models
    (1 warm-up + `runs` timed passes, median) -> clips/s through the whole ensemble."""
    from oracle import ref_torch as oref
    cores = physical_cores()
    torch.set_num_threads(cores)
    folds = w["inference"]["folds"]
    models = []
    for fold in range(folds):
        torch.manual_seed(100 + fold)
        models.append(oref.TagCNN2d(w["features"], w["blocks"], w["base"], w["growth"], w["start"], 80,
                                    output_dropout=w["dropout"]).eval())
    signal, _ = synthetic_batch(w, batch, torch.device("cpu"), 99)
    times = []
    with torch.no_grad():
        for i in range(runs + 1):
            t0 = time.perf_counter()
            acc = None
            for m in models:
                p = torch.sigmoid(m(signal)["class_logits"])
                acc = p if acc is None else acc + p
            times.append(time.perf_counter() - t0)
    times = sorted(times[1:])
    med = times[len(times) // 2]
    return dict(value=batch / med, unit="clips/s", cores=cores, kind="port",
                sample="oracle eval-mode forward of %d x 10 s clips through %d fold models (sigmoid mean), 1 warm-up + %d timed "
                       "passes (median)" % (batch, folds, runs))


def run_other_workloads():
    """cfg 3 and cfg 5 as short runs of this script in fresh processes (N = 1), so that the driver's default invocation also observes
    them: their one-line results, trimmed, go into `other_workloads` of the cfg-2 line.  ~1 minute."""
    import subprocess
    out = {}
    for name, extra in (("cfg3", ["--steps", "20", "--warmup", "5", "--graph", "--no-cpu-baseline"]), ("cfg5", ["--warmup", "2", "--no-alt"])):
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", name, "--no-other"] + extra
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
            line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                out[name] = {"error": "rc %d: %s" % (r.returncode, r.stderr.decode()[-300:])}
                continue
            d = json.loads(line[-1])
        except Exception as e:                                  # a failed side measurement must not take the cfg-2 line down
            out[name] = {"error": repr(e)[:300]}
            continue
        roof = d.get("roofline") or {}
        out[name] = {"metric": d["metric"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                     "dtype": d["dtype"], "scaling": d["scaling"], "workload": d["config"]["workload"],
                     "roofline": {k: roof.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "launches_per_step",
                                                           "avg_launch_ms", "conv_ms_per_step", "conv_ms_total_per_step", "stages", "streams",
                                                           "note", "algorithmic_bytes_per_step", "conv_algorithmic_bytes_per_step",
                                                           "pass_algorithmic_bytes_per_step", "conv_GBps", "how",
                                                           "mfma_view_of_the_largest_conv", "measured_on", "one_stream_clips_per_s",
                                                           "streams_of_value", "conv_ms_total", "wall_ms") if k in roof}}
        for k in ("final_loss", "audio_seconds_per_s", "abi_calls_per_step", "hip_graph", "cpu_baseline", "arith_bits", "fast_mode", "exact_mode", "act_fold",
                  "f32_mode"):
            if k in d:
                out[name][k] = d[k]
    return out


def run_inference(args, w, device, world, rank):
    """cfg 5: length-grouped fold-ensemble inference (reference predict_2d_cnn.py:72-125, README.md:37).  All fold
    weight sets and all padded batches are resident in HBM before the timed region; batches are dealt round-robin to
    the ranks (no data-path collective; total work is fixed -> strong scaling).  Prints ONE JSON line on rank 0."""
    import random

    import numpy as np

    import predict_2d_cnn as drv
    from freesound_classification_amd import functional as F
    from freesound_classification_amd.networks.classifiers import TwoDimensionalCNNClassificationModel
    from freesound_classification_amd.ops.padding import BucketingSampler

    inf = w["inference"]
    rng = np.random.RandomState(inf["seed"])
    lens = rng.randint(int(inf["min_s"] * w["sr"]), int(inf["max_s"] * w["sr"]), size=inf["clips"])

    class _DS:
        lengths = lens

    random.seed(inf["seed"])
    step = int(inf["bucket_s"] * w["sr"])
    edges = list(range(0, int(lens.max()) + 2 * step, step))
    batches = [list(map(int, b)) for b in BucketingSampler(_DS, w["batch"] * w["samples"], edges)]
    mine = batches[rank::world]
    models = []
    for fold in range(inf["folds"]):
        torch.manual_seed(100 + fold)
        m = TwoDimensionalCNNClassificationModel(make_experiment(w), device=str(device))
        for mod in m.modules():                              # non-trivial running statistics for eval-mode BN
            if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
                mod.running_mean.normal_(0, 0.1)
                mod.running_var.uniform_(0.5, 1.5)
        models.append(m.eval())
    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    padded = []
    for b in mine:                                           # what the collate + H2D hands over: zero-padded to the longest
        t = int(lens[b].max())
        x = torch.zeros(len(b), t, 1, device=device)
        for row, i in enumerate(b):
            x[row, :lens[i], 0] = 0.1 * torch.randn(int(lens[i]), device=device, generator=gen)
        padded.append(x)
    n_steps = len(padded) if args.steps <= 0 or args.steps > len(padded) else args.steps
    # warm-up on the LARGEST batches (the caching allocator then holds every size the timed pass asks for on each of the fold
    # streams: with the first batches only, the timed pass of a fresh process paid its hipMallocs -- 1044 against 1130 - 1150 clips/s)
    n_warm = max(1, min(args.warmup, len(padded)))
    for x in sorted(padded, key=lambda t: -t.numel())[:n_warm]:
        drv.ensemble_batch(models, x)
    timer = None if args.no_kernel_timer else F.KernelTimer()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    clips = 0
    # (the residual units' conv -> BatchNorm -> PReLU run as one launch with calibrated operand scales -- the warm-up batch calibrated
    # them on the two-pass route.  The timed pass is THE PRODUCTION PIPELINE, predict_2d_cnn.ensemble_batches as predict_folds
    # drives it: every batch's probabilities and overflow flags go to the host in one asynchronous copy, the next batch is enqueued
    # before that copy is waited for, and a batch whose flags say a calibrated scale was outgrown is recomputed without the fold
    # inside the timed region)
    def run_batches(xs):
        redone0, last = F.EVAL_RECOMPUTES, None
        for last in drv.ensemble_batches(models, xs):
            pass
        return F.EVAL_RECOMPUTES - redone0, last

    refolded, probs = run_batches(padded[:n_steps])
    clips = sum(x.shape[0] for x in padded[:n_steps])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    # per-kernel attribution: a second pass on ONE stream with a HIP-event pair around every convolution launch (on three
    # streams a launch's event duration includes the share of the chip the other streams' kernels took while it ran)
    one_stream = None
    if timer is not None:
        streams0, drv.FOLD_STREAMS = drv.FOLD_STREAMS, 1
        try:
            drv.ensemble_batch(models, padded[0])
            torch.cuda.synchronize()
            F.TIMER = timer
            t1 = time.perf_counter()
            run_batches(padded[:n_steps])
            torch.cuda.synchronize()
            one_stream = time.perf_counter() - t1
        finally:
            F.TIMER = None
            drv.FOLD_STREAMS = streams0
    fast = exact = None

    def repass(mode):                                       # the same pass in another arithmetic, outside `value`
        mode0 = F.get_conv_arith()
        F.set_conv_arith(mode)
        try:
            drv.ensemble_batch(models, padded[0])             # (calibrates the folded launches of this arithmetic)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_batches(padded[:n_steps])
            torch.cuda.synchronize()
            e1 = time.perf_counter() - t1
            return {"conv_arith": ARITH_LABEL[mode], "arith_bits": ARITH_BITS[mode], "value": clips / e1, "unit": "clips/s",
                    "steps": n_steps, "ms_per_step": 1e3 * e1 / n_steps}
        finally:
            F.set_conv_arith(mode0)

    if world == 1 and F.get_conv_arith() != 3:              # the opt-in fast arithmetic (22-bit products)
        fast = repass(3)
    if world == 1 and F.get_conv_arith() == 10:             # exact products (bf16x9)
        exact = repass(9)
    total = torch.tensor([float(clips), elapsed], device=device, dtype=torch.float64)
    if world > 1:
        both = total.clone()
        dist.all_reduce(both[:1], op=dist.ReduceOp.SUM)
        dist.all_reduce(total[1:], op=dist.ReduceOp.MAX)
        total[0] = both[0]
    if not torch.isfinite(probs).all():
        raise SystemExit("non-finite probabilities in the inference benchmark")
    if rank == 0:
        clips_all, tmax = float(total[0]), float(total[1])
        seconds = float(lens.sum()) / w["sr"]
        result = {
            "metric": "inference clips/s (5-fold ensemble, length-grouped batches, GPU STFT+mel)",
            "value": clips_all / tmax, "unit": "clips/s", "n_gpus": world, "steps": n_steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * tmax / n_steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": {0: "f32", 3: "f32 (f16x3 products)", 6: "f32 (bf16x6 products)",
                      9: "f32 (exact products: 3 bf16 limbs x 9 MFMA products, fp32 accumulate)",
                      10: "f32 (3 scaled fp16 limbs x 6 MFMA products: products to 2^-32, fp32 accumulate)"}.get(F.get_conv_arith(), "f32"),
            "arith_bits": ARITH_BITS[F.get_conv_arith()], "data": "synthetic",
            "config": {"workload": "cfg5: %d clips of U(%.1f, %.0f) s @ %.1f kHz (%.0f s of audio), %d length-grouped batches "
                                   "(bucket edges every %.0f s, <= %d x %.0f s of samples), %d resident fold models of the "
                                   "cfg-2 network, sigmoid mean" % (
                                       inf["clips"], inf["min_s"], inf["max_s"], w["sr"] / 1e3, seconds, len(batches),
                                       inf["bucket_s"], w["batch"], w["samples"] / w["sr"], inf["folds"]),
                       "parallelism": "dp%d (batches round-robin)" % world,
                       "conv_arith": ARITH_LABEL[F.get_conv_arith()]},
            "audio_seconds_per_s": seconds * (clips_all / inf["clips"]) / tmax,
            "reference_claim": "README.md:37: stage-1 test set, 5 folds, 'only 1 minute' (hardware unspecified)",
        }
        if timer is not None:
            summ = timer.summary()
            dom_name, dom = max(summ.items(), key=lambda kv: kv[1]["ms"])
            peak, per, arith = price_kernel(dom_name)
            ach = dom["flops"] / dom["ms"] / 1e9
            fam = {}
            for name, r in summ.items():
                f = fam.setdefault(name.split("<")[0], dict(flops=0.0, ms=0.0))
                f["flops"] += r["flops"]
                f["ms"] += r["ms"]
            result["roofline"] = {"kernel": dom_name, "bound": "mfma", "achieved": ach * per, "peak": peak, "unit": "TFLOP/s",
                                  "frac": ach * per / peak, "traffic": None, "arithmetic": arith, "algorithmic_fp32_tflops": ach,
                                  "launches_per_step": dom["launches"] / n_steps, "avg_launch_ms": dom["ms"] / dom["launches"],
                                  "algorithmic_gflop_per_launch": dom["flops"] / dom["launches"] / 1e9,
                                  "conv_ms_per_step": {k: v["ms"] / n_steps for k, v in fam.items()},
                                  "conv_tflops": {k: v["flops"] / v["ms"] / 1e9 for k, v in fam.items()},
                                  "conv_ms_total": sum(v["ms"] for v in summ.values()), "wall_ms": 1e3 * one_stream,
                                  "measured_on": "a second pass with the fold models on ONE stream (a HIP-event pair per convolution "
                                                 "launch): %.0f clips/s; `value` runs them on %d streams without the per-launch events"
                                                 % (clips / one_stream, drv.FOLD_STREAMS),
                                  "one_stream_clips_per_s": clips / one_stream, "streams_of_value": drv.FOLD_STREAMS}
        if fast is not None:
            result["fast_mode"] = fast
        if exact is not None:
            result["exact_mode"] = exact
        result["act_fold"] = {"enabled": bool(F.EVAL_ACT_FOLD), "batches_recomputed_without_it": refolded,
                              "what": "conv -> eval BatchNorm -> PReLU of the residual units in one launch (fsc_conv_l16_fwd_act)"}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline_inference(w)
        _emit(json.dumps(result))
    if dist.is_initialized():
        dist.destroy_process_group()


XGMI_LINK_GBPS = 153.0            # per link and direction; 7 links per GPU (MI355X_MICROARCH.md / SURVEY section 5)
GRAD_BYTES_CFG2 = 21545583 * 4    # the one exchange of the path: a sum all-reduce of the fp32 gradients (86.18 MB)


def self_launch(n):
    """`python bench.py --gpus N` with N > 1 and no torch.distributed.run environment: start the N ranks ourselves
    (one process per GPU on this node, rendezvous on 127.0.0.1) with the same arguments; rank 0's JSON line is the only
    thing that reaches this process's stdout.  Returns the launcher's exit code."""
    from freesound_classification_amd import parallel
    return parallel.launch_ranks(__file__, sys.argv[1:], n)


def allreduce_probe(device, world, nbytes, iters=10, warm=3):
    """Sum all-reduce of an `nbytes` fp32 buffer (the gradient payload) alone on the wire: ms, algorithm GB/s (payload / time)
    and bus GB/s (2 (n - 1) / n x payload / time: what each GPU sends and receives), next to what xGMI offers."""
    buf = torch.ones(nbytes // 4, device=device, dtype=torch.float32)
    sync = torch.cuda.synchronize if device.type == "cuda" else (lambda: None)
    for _ in range(warm):
        dist.all_reduce(buf)
    sync()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        dist.all_reduce(buf)
    sync()
    dt = (time.perf_counter() - t0) / iters
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    bus = 2.0 * (world - 1) / world * nbytes / dt / 1e9
    return {"payload_bytes": int(nbytes), "ms": 1e3 * dt, "algo_GBps": nbytes / dt / 1e9, "bus_GBps": bus,
            "xgmi_link_GBps": XGMI_LINK_GBPS, "xgmi_links_per_gpu": 7,
            "frac_of_one_link": bus / XGMI_LINK_GBPS, "backend": dist.get_backend()}


def launch_check(args, world, rank, local_rank):
    """--launch-check: the rendezvous, barrier and gradient-sized all-reduce of the N-rank job without the model (backend
    `gloo` runs on CPU: tests/test_parallel_cpu.py drives the self-launcher through it; `nccl` = RCCL measures the
    86.18 MB exchange alone).  Rank 0 prints one JSON line."""
    for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
        os.environ.setdefault(k, v)
    if args.backend == "nccl":
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        dist.init_process_group("nccl", device_id=device)
    else:
        device = torch.device("cpu")
        dist.init_process_group("gloo")
    assert dist.get_world_size() == world and dist.get_rank() == rank
    dist.barrier()
    probe = allreduce_probe(device, world, args.check_bytes, iters=3, warm=1)
    ranks = [None] * world
    dist.all_gather_object(ranks, (rank, local_rank, os.getpid()))
    if rank == 0:
        _emit(json.dumps({"launch_check": True, "n_gpus": world, "ranks": ranks, "allreduce": probe}))
    dist.destroy_process_group()


_RESULT_FD = None


def _emit(line):
    sys.stdout.flush()
    os.write(_RESULT_FD if _RESULT_FD is not None else 1, (line + "\n").encode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra native-fp32-MFMA measurement (alt_f32)")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC passes behind roofline.traffic (two child runs, ~1 min)")
    ap.add_argument("--no-kernel-timer-in-value", action="store_true",
                    help="time `value` without the per-kernel HIP events and take the per-kernel times behind `roofline` from a few extra "
                         "steps after it (launch-bound workloads: cfg 3 issues ~700 kernels per step, an event pair around each makes "
                         "the host the bottleneck: 13.2 instead of 11.5 ms)")
    ap.add_argument("--graph", action="store_true", help="cfg3: replay the step from a recorded HIP graph (CapturedTrainingStep) instead "
                                                         "of issuing it eagerly; the whole run then lives on one non-default stream")
    ap.add_argument("--no-graph", action="store_true", help="(default; kept for the profile scripts)")
    ap.add_argument("--no-other", action="store_true", help="skip the short cfg3 / cfg5 runs attached to the default cfg2 line (other_workloads)")
    ap.add_argument("--kernel-table", action="store_true", help="print per-kernel timing to stderr")
    ap.add_argument("--arith", default=None, choices=["f32", "bf16", "f16x3", "bf16x6", "bf16x9", "f16x6"],
                    help="conv arithmetic (default: the workload's (cfg3: bf16), else the library default = f16x6)")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous + barrier + gradient-sized all-reduce only (no model); with --backend gloo it runs on CPU")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend (gloo: --launch-check only)")
    ap.add_argument("--check-bytes", type=int, default=GRAD_BYTES_CFG2, help="payload of the --launch-check all-reduce")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.backend != "nccl" and not args.launch_check:
        raise SystemExit("--backend gloo is for --launch-check only: the step itself needs the GPUs (no CPU fallback)")

    # N > 1 without a launcher environment: become the launcher (the driver calls `python bench.py --gpus N ...`)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    # stdout carries exactly ONE line, the JSON result: anything the libraries print there (RCCL writes its version banner to
    # stdout when the first communicator comes up) goes to stderr instead
    global _RESULT_FD
    sys.stdout.flush()
    _RESULT_FD = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but the launcher environment says WORLD_SIZE=%d" % (args.gpus, world))
    if args.launch_check:
        return launch_check(args, world, rank, local_rank)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("FSC_FORCE_DP") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)

    from freesound_classification_amd import functional as F
    from freesound_classification_amd.networks.classifiers import (
        HierarchicalCNNClassificationModel, TwoDimensionalCNNClassificationModel)
    from freesound_classification_amd.ops.training import make_step

    w = WORKLOADS[args.workload]
    if args.arith or w.get("arith"):
        F.set_conv_arith(args.arith or w["arith"])
    if "inference" in w:
        if "--steps" not in sys.argv:
            args.steps = 0                      # one pass over every length-grouped batch
        return run_inference(args, w, device, world, rank)
    batch = args.batch or w["batch"]
    if args.graph:
        # CapturedTrainingStep wants eager steps, recording and replays on ONE non-default stream
        run_stream = torch.cuda.Stream(device=device)
        run_stream.wait_stream(torch.cuda.current_stream(device))
        torch.cuda.set_stream(run_stream)
    torch.manual_seed(42)
    model_cls = HierarchicalCNNClassificationModel if w.get("dims") == 1 else TwoDimensionalCNNClassificationModel
    model = model_cls(make_experiment(w), device=str(device))
    model.train()
    model.global_step = 0
    # (every step the run takes must lie inside the one-cycle schedule -- past its end the rate goes negative: warm-up, the timed steps,
    # the extra measurements behind them, two replays after a graph capture)
    model.make_optimizer(max_steps=args.steps + args.warmup + 24)
    signal, labels = synthetic_batch(w, batch, device, 1234 + rank)

    mix_rng = __import__("numpy").random.RandomState(7 + rank)

    step_fn = [model.training_step]

    def one_step():
        model.global_step += 1
        make_step(model.scheduler, step=model.global_step)
        x, y = signal, labels
        if w.get("mixup"):
            # on-device MixUp (ops/audio.py:32-52 semantics): partner = a batch permutation, rows are
            # mixed with probability p; equal lengths -> plain average, labels OR-ed
            # (partner table: rows that do not draw MixUp pass through, no gathered copy of the partner rows)
            perm = mix_rng.permutation(batch)
            take = mix_rng.uniform(size=batch) < w["mixup"]
            partner = __import__("numpy").where(take, perm, -1)
            t = w["samples"]
            mixed, y = F.mixup_rows(signal.squeeze(-1), signal.squeeze(-1), partner, [t] * batch, [t] * batch, [0] * batch,
                                    mix_rng.uniform(0.4, 0.6, size=batch), labels, labels)
            x = mixed.unsqueeze(-1)
        out = step_fn[0](x, y)
        if os.environ.get("FSC_BENCH_SYNC_ALL"):
            torch.cuda.synchronize()
            print("step %d loss %.3f max|logit| %.1f" % (model.global_step, float(out[2].detach()), float(out[0].detach().abs().max())), file=sys.stderr)
        return out

    for _ in range(args.warmup):
        one_step()
    # --graph (cfg 3): the step is recorded once as a HIP graph and replayed (ops/training.py CapturedTrainingStep: same entry points,
    # arguments and order; learning rate / step count through device memory; MixUp stays outside, its draws are host-side).  Off by
    # default: this host issues the 510 entry-point calls of a cfg-3 step in 9 ms against 11.5 ms of kernels, so the replay
    # measures the same 11.6 ms -- it pays on a slower host or a faster GPU.  The per-kernel times behind `roofline` then come
    # from a few eager steps AFTER the timed region.
    use_graph = (args.graph and w.get("dims") == 1 and float(w["dropout"]) == 0.0 and world == 1
                 and os.environ.get("FSC_FORCE_DP") != "1")
    captured = None
    if use_graph:
        from freesound_classification_amd.ops.training import CapturedTrainingStep
        # FSC_GRAPH_ASYNC_WGRAD=1: weight gradients on a forked stream -- in the recorded graph they become branches beside the
        # BatchNorm / dgrad chain, so that the small late-block kernels could overlap instead of queueing
        F.ASYNC_WGRAD = os.environ.get("FSC_GRAPH_ASYNC_WGRAD", "0") == "1"      # (measured: 11.71 against 11.53 - 11.60 ms, not a gain)
        try:
            captured = CapturedTrainingStep(model, signal, labels)
        finally:
            F.ASYNC_WGRAD = False
        step_fn[0] = captured
        for _ in range(2):
            one_step()
    post_timer = (use_graph or args.no_kernel_timer_in_value) and not args.no_kernel_timer
    timer = None
    if not args.no_kernel_timer and not post_timer:
        timer = F.KernelTimer()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    F.TIMER = timer
    calls0 = F._lib.CALLS[0]
    t0 = time.perf_counter()
    trace = [] if os.environ.get("FSC_BENCH_TRACE_LOSS") else None      # development: the loss of every timed step (synchronises each)
    for _ in range(args.steps):
        logits, per, loss = one_step()
        if trace is not None:
            lg_ = logits.detach()
            trace.append((float(loss.detach()), float(lg_.abs().max()), float((lg_.max(dim=1).values - lg_.min(dim=1).values).max()),
                          float(model.optimizer.param_groups[0]["lr"])))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if trace is not None:
        print("timed steps, loss / max |logit| / largest score gap / lr:", " ".join("%.3f/%.0f/%.0f/%.5f" % v for v in trace), file=sys.stderr)
    F.TIMER = None
    abi_calls = (F._lib.CALLS[0] - calls0) / max(1, args.steps)
    timer_steps = args.steps
    graph_info = None
    if use_graph:
        captured.sync_state()
        step_fn[0] = model.training_step
        graph_info = {"replayed": True, "abi_calls_per_replayed_step": abi_calls}
    if post_timer:                                       # per-kernel attribution: eager steps with the timer, outside the timed region
        one_step()
        torch.cuda.synchronize()
        timer = F.KernelTimer()
        timer_steps = 5
        F.TIMER = timer
        calls0 = F._lib.CALLS[0]
        t_e = time.perf_counter()
        for _ in range(timer_steps):
            one_step()
        torch.cuda.synchronize()
        timed_ms = 1e3 * (time.perf_counter() - t_e) / timer_steps
        F.TIMER = None
        abi_calls = (F._lib.CALLS[0] - calls0) / timer_steps
        if graph_info is not None:
            graph_info["eager_ms_per_step_with_kernel_timer"] = timed_ms
    per_rank = None
    exchange = None
    if world > 1:
        mine = torch.tensor([elapsed], device=device, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [batch * args.steps / float(t.item()) for t in every]          # clips/s of each rank over its own clock
        elapsed = max(float(t.item()) for t in every)
        # the exchange step alone (outside the timed region): the gradient all-reduce with nothing else on the wire
        exchange = allreduce_probe(device, world, 4 * sum(p.numel() for p in model.parameters()))
    final_loss = float(loss.detach())
    # The same step with the batch arriving from the host: the 4 * T * N bytes (226 MB at cfg 2) are copied from a pinned
    # staging buffer on a copy stream into one of two device buffers while the previous step computes (double
    # buffering, as ops/device_pipeline.py does for real loaders).  N = 1 only, outside the timed region of `value`.
    # (Measured before the native-fp32 re-run below: straight after those steps the same loop runs 5-11 ms per step slower -- 2750-3100
    # clips/s -- although a resident step does not; not reproduced outside bench.py, tools/h2d_probe.py.)
    h2d = None
    if world == 1 and not args.no_alt and not w.get("mixup"):
        pinned = signal.cpu().pin_memory()
        dev_buf = [torch.empty_like(signal), torch.empty_like(signal)]
        copy_stream = torch.cuda.Stream(device=device)
        events = [None, None]

        def upload(slot):
            copy_stream.wait_stream(torch.cuda.current_stream(device))      # the buffer's previous consumer is done
            with torch.cuda.stream(copy_stream):
                dev_buf[slot].copy_(pinned, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            events[slot] = ev

        def h2d_step(k):
            slot = k & 1
            torch.cuda.current_stream(device).wait_event(events[slot])
            upload(slot ^ 1)                                                # next batch on the wire during this step
            model.global_step += 1
            make_step(model.scheduler, step=model.global_step)
            return model.training_step(dev_buf[slot], labels)

        upload(0)
        for k in range(2):
            h2d_step(k)
        torch.cuda.synchronize()
        k_h = max(2, min(args.steps, 10))
        t2 = time.perf_counter()
        for k in range(k_h):
            h2d_step(k)
        torch.cuda.synchronize()
        e_h = time.perf_counter() - t2
        h2d = {"value": batch * k_h / e_h, "unit": "clips/s", "steps": k_h, "ms_per_step": 1e3 * e_h / k_h,
               "h2d_bytes_per_step": pinned.numel() * 4,
               "note": "host->device copy of the waveform batch inside the timed region (pinned, double-buffered, copy stream)"}
    # The same workload in the other arithmetics, N = 1 only, outside the timed region of `value`: the library's shipped default
    # (f16x3: two scaled fp16 limbs, three products -- the FAST mode, 22-bit products) with its own kernel table, and the native
    # fp32-MFMA kernels (v_mfma_f32_16x16x4_f32).
    def retime(mode, k, warm=2):
        mode0 = F.get_conv_arith()
        F.set_conv_arith(mode)
        try:
            for _ in range(warm):
                one_step()
            torch.cuda.synchronize()
            tm = F.KernelTimer() if timer is not None else None      # same per-call event overhead as the timed region
            F.TIMER = tm
            t1 = time.perf_counter()
            for _ in range(k):
                one_step()
            torch.cuda.synchronize()
            e = time.perf_counter() - t1
        finally:
            F.TIMER = None
            F.set_conv_arith(mode0)
        r = {"conv_arith": ARITH_LABEL[mode], "arith_bits": ARITH_BITS[mode], "value": batch * k / e, "unit": "clips/s", "steps": k,
             "ms_per_step": 1e3 * e / k}
        if tm is not None:
            r["roofline"] = roofline_of(tm.summary(), k)
        return r

    alt = fast = exact = None
    if world == 1 and not args.no_alt and F.get_conv_arith() not in (0, 1):
        if F.get_conv_arith() != 3:
            fast = retime(3, max(2, min(args.steps, 10)))
        if F.get_conv_arith() == 10:
            exact = retime(9, max(2, min(args.steps, 10)))
        alt = retime(0, max(2, min(args.steps, 5)))
    f32_mode = None
    if world == 1 and not args.no_alt and F.get_conv_arith() == 1:
        # a bf16 workload (cfg 3) re-timed at the reference's fp32 precision -- the library default arithmetic; the 1-d model has
        # no pre-split (L16) route, so its convolutions run the fp32-input kernels with exact products (three bf16 limbs, nine
        # MFMA products): the number VERDICT r5 item 5 asked to see on the driver line
        f32_mode = retime(F._lib.load().fsc_conv_default_arith(), max(2, min(args.steps, 5)))
        f32_mode["note"] = ("the same step at the library default arithmetic (f16x6 on the layers with a pre-split tiling, fp32-exact "
                            "nine-product kernels on the others), eager: host-bound on boxes with a slow host")
        if use_graph:
            # ... and replayed from a HIP graph like `value` (the eager figure above depends on the box's host: 13 - 17 ms)
            from freesound_classification_amd.ops.training import CapturedTrainingStep
            mode0 = F.get_conv_arith()
            F.set_conv_arith(F._lib.load().fsc_conv_default_arith())
            try:
                one_step()
                cap2 = CapturedTrainingStep(model, signal, labels)
                step_fn[0] = cap2
                for _ in range(2):
                    one_step()
                torch.cuda.synchronize()
                k2 = max(2, min(args.steps, 10))
                t1 = time.perf_counter()
                for _ in range(k2):
                    one_step()
                torch.cuda.synchronize()
                e2 = time.perf_counter() - t1
                cap2.sync_state()
                f32_mode["eager"] = {"value": f32_mode["value"], "ms_per_step": f32_mode["ms_per_step"]}
                f32_mode["value"], f32_mode["ms_per_step"], f32_mode["steps"] = batch * k2 / e2, 1e3 * e2 / k2, k2
                f32_mode["hip_graph"] = True
            finally:
                step_fn[0] = model.training_step
                F.set_conv_arith(mode0)
    if not torch.isfinite(torch.tensor(final_loss)):
        raise SystemExit("non-finite loss in the benchmark: %r" % final_loss)
    # The HBM-bound stages of BASELINE.md section 3 (front-end, BatchNorm / PReLU / pooling passes, optimizer): a few extra steps
    # with a HIP event pair around every such call (outside the timed region of `value`: ~250 event pairs per step), algorithmic
    # bytes of the call over its time against the 8.0 TB/s HBM peak.  N = 1 only.
    stages = None
    if world == 1 and not args.no_alt and timer is not None:
        F.STAGE_TIMER = F.StageTimer()
        k_s = 3
        for _ in range(k_s):
            one_step()
        torch.cuda.synchronize()
        if os.environ.get("FSC_STAGE_DUMP"):           # development: every call of the last step, (stage, MB, us)
            rows = [(nm, nb / 1e6, 1e3 * e0.elapsed_time(e1)) for nm, nb, e0, e1 in F.STAGE_TIMER.records]
            with open(os.environ["FSC_STAGE_DUMP"], "w") as fh:
                for nm, mb, us in rows[-(len(rows) // k_s):]:
                    fh.write("%-12s %10.1f MB %9.1f us %7.2f TB/s\n" % (nm, mb, us, mb / max(us, 1e-3)))
        # the same per stage over its LARGE calls only (>= 256 MB of algorithmic bytes: beyond the Infinity Cache): a stage's overall
        # figure mixes them with the small late-block calls, which are launch- and latency-bound whatever the kernel does
        large = {}
        for nm, nb, e0, e1 in F.STAGE_TIMER.records:
            if nb >= 256e6:
                r = large.setdefault(nm, [0.0, 0.0, 0])
                r[0] += nb
                r[1] += e0.elapsed_time(e1)
                r[2] += 1
        summ_s, F.STAGE_TIMER = F.STAGE_TIMER.summary(), None
        names = {"frontend": "front-end: waveform -> log-mel / log-STFT (reads 4 T, writes 4 F frames per clip)",
                 "bn_stats": "BatchNorm statistics passes that are not folded into a producer (1 read)",
                 "bn_act_fwd": "BatchNorm + PReLU (+ residual) apply passes, forward (reads + writes counted once each)",
                 "bn_act_bwd": "BatchNorm + PReLU backward: reduce pass + apply pass (+ un-pooling), both passes' reads and writes",
                 "pool": "stand-alone max-pool / global max-pool passes", "optimizer": "Adam-amsgrad, nine fp32 streams"}
        stages = {}
        for k, r in summ_s.items():
            if r["bytes"] <= 0 or r["ms"] <= 0:
                continue
            tbps = r["bytes"] / r["ms"] / 1e9
            stages[k] = {"what": names.get(k, k), "bound": "hbm", "achieved": tbps * 1e3, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                         "frac": tbps * 1e3 / PEAK_HBM_GBPS, "ms_per_step": r["ms"] / k_s, "GB_per_step": r["bytes"] / k_s / 1e9,
                         "calls_per_step": r["calls"] / k_s}
            if k in large and large[k][1] > 0:
                lb, lms, lc = large[k]
                stages[k]["large_calls"] = {"min_bytes": 256e6, "calls_per_step": lc / k_s, "ms_per_step": lms / k_s,
                                            "achieved": lb / lms / 1e6, "frac": lb / lms / 1e6 / PEAK_HBM_GBPS}

    if rank == 0:
        result = {
            "metric": "training-step clips/s (STFT->mel->CNN->LSEP)",
            "value": world * batch * args.steps / elapsed,
            "unit": "clips/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            # f32: tensors, accumulators and results are fp32 (see roofline.arithmetic for how the conv products are formed);
            # bf16: conv operands rounded to bf16, fp32 accumulation / storage / master weights
            # (f16x3 products: every fp32 product is formed from two scaled fp16 limbs per operand, low x low dropped)
            "dtype": {0: "f32", 1: "bf16", 3: "f32 (f16x3 products)", 6: "f32 (bf16x6 products)",
                      9: "f32 (exact products: 3 bf16 limbs x 9 MFMA products, fp32 accumulate)",
                      10: "f32 (3 scaled fp16 limbs x 6 MFMA products: products to 2^-32, fp32 accumulate)"}[F.get_conv_arith()],
            "arith_bits": ARITH_BITS[F.get_conv_arith()],   # significand bits a conv product keeps (24: the exact fp32 product)
            "data": "synthetic",
            "config": {"workload": "%s: batch %d x %.0f s @ %.1f kHz, %s, %d-block %dd CNN base %d growth %g, "
                                   "LSEP, Adam-amsgrad, dropout %g" % (
                                       args.workload, batch, w["samples"] / w["sr"], w["sr"] / 1e3, w["features"],
                                       w["blocks"], w.get("dims", 2), w["base"], w["growth"], w["dropout"]),
                       "global_batch": world * batch, "parallelism": "dp%d" % world,
                       "conv_arith": ARITH_LABEL[F.get_conv_arith()]},
            "final_loss": final_loss,
            "abi_calls_per_step": abi_calls,           # entry-point calls of libfsc_hip.so per (eager) step (each enqueues one to three kernels)
        }
        if post_timer:
            result["kernel_timer"] = "separate pass: %d eager steps with a HIP event pair around every conv launch, after the timed region" % timer_steps
        if graph_info is not None:
            result["hip_graph"] = graph_info
        if timer is not None:
            summ = timer.summary()
            fam = {}
            for name, r in summ.items():
                key = name.split("<")[0]
                f = fam.setdefault(key, dict(launches=0, flops=0.0, ms=0.0))
                for k in f:
                    f[k] += r[k]
            if args.kernel_table:
                for name, r in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
                    print("%-36s launches %5d  ms/step %8.3f  TFLOP/s %7.2f" % (
                        name, r["launches"], r["ms"] / timer_steps, r["flops"] / r["ms"] / 1e9), file=sys.stderr)
            dom_name = max(summ.items(), key=lambda kv: kv[1]["ms"])[0]
            result["roofline"] = roofline_of(summ, timer_steps)
            peak, executed_per_flop, arith = price_kernel(dom_name)
            achieved = result["roofline"]["algorithmic_fp32_tflops"]
            if world == 1 and not args.no_alt and not args.no_traffic:
                tb, note = measure_traffic(dom_name, ["--workload", args.workload, "--arith", ARITH_LABEL[F.get_conv_arith()]]
                                           + (["--batch", str(args.batch)] if args.batch else []))
                result["roofline"]["traffic"] = tb
                result["roofline"]["traffic_source"] = note
            if dom_name in timer.bytes and timer.bytes[dom_name][1]:
                # algorithmic bytes of the same launches: both operands once + the result (three-limb operands 6 B, fp32 4 B per element)
                ab = timer.bytes[dom_name][0] / timer.bytes[dom_name][1]
                result["roofline"]["algorithmic_bytes_per_launch"] = ab
                if result["roofline"].get("traffic"):
                    result["roofline"]["traffic_over_algorithmic"] = result["roofline"]["traffic"] / ab
            if stages:
                result["roofline"]["stages"] = stages
            if w.get("dims") == 1 and stages:
                # cfg 3 is HBM- / launch-bound, not matrix-bound (its largest convolution runs 48 us; a fraction of the MFMA peak on
                # that kernel says nothing): the step is priced against HBM -- ALGORITHMIC bytes of everything a step calls
                # (every convolution's operands once + result once, fp32 tensors; every BatchNorm / PReLU / pool / front-end /
                # optimizer pass by the tensors it reads and writes) over the step time of `value`.
                conv_b = sum(v[0] for v in timer.bytes.values()) / timer_steps
                stage_b = sum(v["GB_per_step"] for v in stages.values()) * 1e9
                step_s = 1e-3 * result["ms_per_step"]
                gbps = (conv_b + stage_b) / step_s / 1e9
                mfma_view = {k: result["roofline"][k] for k in ("kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms",
                                                                "launches_per_step", "arithmetic") if k in result["roofline"]}
                conv_ms = result["roofline"].get("conv_ms_total_per_step")
                result["roofline"].update({
                    "bound": "hbm", "kernel": "whole step (%d entry-point calls)" % round(abi_calls), "achieved": gbps, "peak": PEAK_HBM_GBPS,
                    "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS, "traffic": None,
                    "algorithmic_bytes_per_step": conv_b + stage_b, "conv_algorithmic_bytes_per_step": conv_b,
                    "pass_algorithmic_bytes_per_step": stage_b,
                    "conv_GBps": (conv_b / (1e-3 * conv_ms) / 1e9) if conv_ms else None,
                    "how": "sum of the algorithmic bytes of every call of a step / ms_per_step of `value`; per-stage rates in `stages`, "
                           "the convolutions' own rate in conv_GBps (bytes over the sum of their HIP-event times)",
                    "mfma_view_of_the_largest_conv": mfma_view})
            # The clock the chip actually sustains under the dominant kernel (its largest layer re-run back to back, outside the
            # timed region; the kernel stamps the shader-cycle and the 100 MHz reference counters itself): `peak` above is quoted at
            # the 2.4 GHz maximum, the MFMA-bound launches of this workload run power-limited well below it.
            if world == 1 and dom_name in timer.shapes and ("conv_l16_" in dom_name or "conv_l3_" in dom_name):
                _fl, shape, kind = timer.shapes[dom_name]
                mhz = F.measure_l16_clock(shape, kind)
                if mhz > 0:
                    result["roofline"]["clock"] = {
                        "shader_mhz": mhz, "max_mhz": 2400.0, "layer": list(shape), "kind": kind,
                        "peak_at_clock": peak * mhz / 2400.0, "frac_at_clock": achieved * executed_per_flop / (peak * mhz / 2400.0),
                        "how": "s_memtime / s_memrealtime stamped by workgroup 0 of the kernel, 40 launches of the layer back to back"}
        if per_rank is not None:
            result["per_rank_clips_per_s"] = per_rank
            result["allreduce"] = exchange
        if fast is not None:
            result["fast_mode"] = fast                                      # the shipped default arithmetic (22-bit products): an extra, not `value`
            result["config"]["fast_mode_f16x3_clips_per_s"] = fast["value"]
        if exact is not None:
            result["exact_mode"] = exact                                    # exact products whatever the operands' range (bf16x9): an extra, not `value`
            result["config"]["exact_mode_bf16x9_clips_per_s"] = exact["value"]
        if f32_mode is not None:
            result["f32_mode"] = f32_mode
        if alt is not None:
            result["alt_f32"] = alt
            result["config"]["native_f32_mfma_clips_per_s"] = alt["value"]  # the same step on the native fp32-MFMA kernels
        if h2d is not None:
            result["with_h2d"] = h2d
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(w)
        if world == 1 and args.workload == "cfg2" and not args.no_other and not args.no_alt and args.batch is None:
            del model, signal, logits, per, loss               # the side runs get the whole GPU
            F.forget_packed_weights()
            torch.cuda.empty_cache()
            result["other_workloads"] = run_other_workloads()
        # the numbers of the line once more, compact and LAST (a log tail keeps them)
        ow = result.get("other_workloads") or {}
        roof = result.get("roofline") or {}
        result["summary"] = {
            "cfg2_clips_per_s": result["value"], "cfg2_ms_per_step": result["ms_per_step"], "cfg2_arith": result["config"]["conv_arith"],
            "cfg2_arith_bits": result["arith_bits"], "cfg2_roofline_kernel": roof.get("kernel"), "cfg2_roofline_frac": roof.get("frac"),
            "cfg2_fast_mode_f16x3_clips_per_s": fast["value"] if fast else None,
            "cfg2_exact_mode_bf16x9_clips_per_s": exact["value"] if exact else None,
            "cfg2_native_f32_mfma_clips_per_s": alt["value"] if alt else None,
            "cfg3_clips_per_s": (ow.get("cfg3") or {}).get("value"), "cfg3_ms_per_step": (ow.get("cfg3") or {}).get("ms_per_step"),
            "cfg3_hbm_frac": ((ow.get("cfg3") or {}).get("roofline") or {}).get("frac"),
            "cfg3_f32_mode_clips_per_s": ((ow.get("cfg3") or {}).get("f32_mode") or {}).get("value"),
            "cfg5_clips_per_s": (ow.get("cfg5") or {}).get("value"),
            "cfg5_fast_mode_f16x3_clips_per_s": ((ow.get("cfg5") or {}).get("fast_mode") or {}).get("value"),
            "cfg5_exact_mode_bf16x9_clips_per_s": ((ow.get("cfg5") or {}).get("exact_mode") or {}).get("value"),
            "cpu_baseline_clips_per_s": (result.get("cpu_baseline") or {}).get("value")}
        _emit(json.dumps(result))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
