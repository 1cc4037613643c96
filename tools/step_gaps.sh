#!/bin/bash
# GPU idle time inside one bench step: kernel-trace span vs. the sum of kernel durations (development tool).
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout -k 5 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ktrace -o p -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-alt "$@" > $R/gpurun_out/ktrace.log 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ktrace/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# steps are delimited by the front-end kernel
starts = [i for i, r in enumerate(rows) if "frontend" in r[2]]
print("front-end launches:", len(starts), "kernels:", len(rows))
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a:b]
    span = seg[-1][1] - seg[0][0]
    busy = 0
    cur_s, cur_e = seg[0][0], seg[0][1]
    for s, e, _ in seg[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    gaps = sorted(((seg[i + 1][0] - max(x[1] for x in seg[:i + 1][-8:]), seg[i][2][:50], seg[i + 1][2][:50]) for i in range(len(seg) - 1)), reverse=True)
    print("step: %d kernels, span %.3f ms, busy (union) %.3f ms, idle %.3f ms, sum of durations %.3f ms"
          % (len(seg), span / 1e6, busy / 1e6, (span - busy) / 1e6, sum(e - s for s, e, _ in seg) / 1e6))
    for g, k0, k1 in gaps[:12]:
        print("   gap %7.1f us  after %-50s before %s" % (g / 1e3, k0, k1))
PY
rm -rf gpurun_out/ktrace
