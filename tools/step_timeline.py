"""Development: the kernel launches of the LAST training step in a rocprofv3 kernel trace, in stream order, with each kernel's
duration and the idle gap in front of it -- where a step's time goes between the big kernels.
    python tools/step_timeline.py <kernel_trace.csv> <steps traced> > timeline.txt"""
import csv
import re
import sys

path, steps = sys.argv[1], int(sys.argv[2])
rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# a step begins at its front-end kernel
starts = [i for i, r in enumerate(rows) if "frontend" in r[2]]
begin = starts[-1] if starts else len(rows) - len(rows) // steps
last = rows[begin:]
short = lambda s: re.sub(r"\(.*", "", re.sub(r"^void |\(anonymous namespace\)::|fsc::|at::native::", "", s))[:70]
busy = gap = 0
prev_end = last[0][0]
print("# %d launches, kernel / us / gap in front (us)" % len(last))
agg = {}
for s, e, name in last:
    g = max(0, s - prev_end)
    print("%-72s %9.1f %7.1f" % (short(name), (e - s) / 1e3, g / 1e3))
    busy += e - s
    gap += g
    a = agg.setdefault(short(name), [0, 0.0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3; a[2] += g / 1e3
    prev_end = max(prev_end, e)
print("# step: %.3f ms busy, %.3f ms idle between kernels, %.3f ms wall" % (busy / 1e6, gap / 1e6, (prev_end - last[0][0]) / 1e6))
print("# per kernel: launches, total us, total gap in front")
for k, (n, us, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("#   %-70s %4d %9.1f %8.1f" % (k, n, us, g))
