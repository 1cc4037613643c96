#!/bin/bash
# HBM traffic of one tools/l3_check.py invocation per kernel (development).  usage: tools/pmc_traffic.sh TAG <l3_check arguments>
# FETCH_SIZE counts 64 B per 128-byte request on gfx950 (x2, MI355X guide), WRITE_SIZE x1; both in KB.
TAG=$1; shift
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/$TAG/$c -o p -- python $R/tools/l3_check.py --iters 2 "$@" > $R/gpurun_out/$TAG.$c.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/$TAG/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_l3_fwd" not in k and "wgrad_kernel" not in k: continue
        agg[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    f = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1) * 1024 * 2
    w = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1) * 1024
    print("%-90s fetch %.1f MB  write %.1f MB  total %.1f MB (%d launches)" % (k, f / 1e6, w / 1e6, (f + w) / 1e6, len(d["FETCH_SIZE"])))
PY
