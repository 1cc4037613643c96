#!/bin/bash
# Step anatomy of the data-parallel path in a 1-rank process group (development tool): kernel-trace of bench.py with FSC_FORCE_DP=1
R=$PWD; cd /tmp && export TMPDIR=/tmp
FSC_FORCE_DP=1 timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dptrace -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-alt > $R/gpurun_out/dptrace.log 2>&1
cd $R
tail -2 gpurun_out/dptrace.log | cut -c1-300
f=$(find gpurun_out/dptrace -name "*kernel_stats.csv" | head -1); python tools/summarize_kernel_stats.py $f 5 "dp" | head -30
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/dptrace/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
starts = [i for i, r in enumerate(rows) if "frontend" in r[2]]
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a:b]
    span = seg[-1][1] - seg[0][0]
    busy = 0; cur_s, cur_e = seg[0][0], seg[0][1]
    gaps = []
    for s, e, n in seg[1:]:
        if s > cur_e:
            gaps.append((s - cur_e, n[:60]))
            busy += cur_e - cur_s; cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    gaps.sort(reverse=True)
    print("step: %d kernels, span %.3f ms, busy %.3f ms, idle %.3f ms" % (len(seg), span / 1e6, busy / 1e6, (span - busy) / 1e6))
    for g, n in gaps[:8]:
        print("   gap %8.1f us before %s" % (g / 1e3, n))
PY
rm -rf gpurun_out/dptrace
