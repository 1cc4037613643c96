"""Development: per-kernel totals of a tools/step_timeline.py listing (calls, us per step), largest first.   python tools/tl_agg.py <timeline.txt>"""
import re
import sys
from collections import defaultdict

tot, cnt = defaultdict(float), defaultdict(int)
busy = gaps = 0.0
for line in open(sys.argv[1]):
    m = re.match(r"(.*?)\s+([\d.]+)\s+([\d.]+)$", line.rstrip())
    if not m or line.startswith("#"):
        continue
    tot[m.group(1)] += float(m.group(2)); cnt[m.group(1)] += 1
    busy += float(m.group(2)); gaps += float(m.group(3))
for name, t in sorted(tot.items(), key=lambda kv: -kv[1])[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print("%-64s %4d %9.1f" % (name, cnt[name], t))
print("busy %.1f us, gaps %.1f us" % (busy, gaps))
