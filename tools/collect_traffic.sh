#!/bin/bash
# HBM bytes per launch of every conv kernel in the bench step, from rocprofv3 PMC counters collected in
# separate passes with --kernel-trace only (MI355X guide, HBM section): FETCH_SIZE (x2: gfx950 reports
# half, calibrated on fsc_axpy in tools/traffic_probe.py) and WRITE_SIZE (x1), both in KiB.
# Writes profiles/roofline_traffic.json (read by bench.py for roofline.traffic).  Run on the GPU box.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/traffic/$c -o p -- \
      python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-alt --no-other > $R/gpurun_out/traffic_$c.log 2>&1
done
python - <<PY
import csv, glob, json, re, collections
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for c, mult in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
    for f in glob.glob("$R/gpurun_out/traffic/%s/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "conv_" not in n: continue
            m = re.search(r"(conv_\w+)<([^>]*)>", n)
            if not m: continue
            base, args = m.group(1), [a.strip() for a in m.group(2).split(",")]
            if base == "conv_fwd_kernel": args = args[:4]                       # timer name drops the K-chunk argument
            if base == "conv_l16_fwd_kernel": args = args[:4] + (["pool"] if args[4:5] == ["true"] else [])   # (<KH,KW,COT,PT,POOL,STATS>)
            if base == "conv_wgrad_kernel": args = args[:3] + (["packed"] if args[3] == "true" else [])
            key = "%s<%s>" % (base, ",".join(args))
            tot[key] += float(r["Counter_Value"]) * 1024.0 * mult
            if c == "FETCH_SIZE": cnt[key] += 1
out = {"_note": "HBM bytes per launch from rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes over bench.py --steps 2 --warmup 1), "
               "averaged over all launches of the kernel; FETCH_SIZE x2 (gfx950 half-count, calibrated on fsc_axpy: 1048579 KiB reported for "
               "2 GiB read), WRITE_SIZE x1.  Regenerate with tools/collect_traffic.sh."}
for k in sorted(tot): out[k] = int(tot[k] / max(cnt[k], 1))
json.dump(out, open("$R/gpurun_out/roofline_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
