"""rocprofv3 --kernel-trace --stats CSV -> the per-step table kept under profiles/ (development tool).

    python tools/summarize_kernel_stats.py <p_kernel_stats.csv> <steps traced> "<header line>" ...
"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for h in sys.argv[3:]:
    print("# " + h)
print("# columns: kernel, calls per step, ms per step, average us per launch, percent of GPU time\n")


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:66]


for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    c, t = int(r["Calls"]), float(r["TotalDurationNs"])
    print("%-68s %6.1f %9.3f %10.1f %6.2f" % (short(r["Name"]), c / steps, t / steps / 1e6, t / c / 1e3, 100 * t / tot))
print("\ntotal GPU kernel time: %.1f ms/step" % (tot / steps / 1e6))
