#!/bin/bash
# Per-launch durations of selected kernels over one bench run (development tool).  usage: tools/trace_kernels.sh <regex> [bench args]
pat=$1; shift
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout -k 5 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ktrace -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-alt "$@" > $R/gpurun_out/ktrace.log 2>&1
cd $R
python - "$pat" <<'PY'
import csv, glob, re, sys
pat = re.compile(sys.argv[1])
f = glob.glob("gpurun_out/ktrace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
half = len(rows) // 2
for r in rows[half:]:
    n = r["Kernel_Name"]
    if pat.search(n):
        print("%-60s grid %-12s %8.1f us" % (re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))[:60],
                                            r.get("Grid_Size", "?"), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
rm -rf gpurun_out/ktrace
