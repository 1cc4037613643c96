#!/bin/bash
# One PMC pass over the whole bench step, aggregated per kernel (development tool).  usage: tools/pmc_bench.sh "<counters>" <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 5 400 rocprofv3 --kernel-trace --pmc $1 --output-format csv -d $R/gpurun_out/$2 -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-alt > $R/gpurun_out/$2.log 2>&1
python - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$R/gpurun_out/$2/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        n = re.sub(r"\(.*", "", n)[:48]
        agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
names = sorted({c for k in agg for c in agg[k]})
print("%-50s" % "kernel (sum over 3 steps)" + "".join("%22s" % c for c in names))
for k in sorted(agg, key=lambda k: -sum(agg[k].values()))[:28]:
    print("%-50s" % k + "".join("%22.4g" % agg[k][c] for c in names))
PY
