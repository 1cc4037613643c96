"""Development: per kernel of a rocprofv3 kernel trace (CSV), the launch durations of the LAST traced step sorted -- shows the floor a
small launch pays.   python tools/trace_hist.py <p_kernel_trace.csv> <launches per step of the kernel family> [name filter ...]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
filt = sys.argv[2:]
by = defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if filt and not any(f in n for f in filt):
        continue
    by[n].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
for n, v in sorted(by.items(), key=lambda kv: -sum(d for _, d in kv[1])):
    v.sort()
    per = len(v) // 7 if len(v) >= 7 else len(v)
    last = sorted(d for _, d in v[-per:])
    print("%-40s %3d per step, us: %s" % (n[:40], per, " ".join("%.0f" % d for d in last)))
