#!/bin/bash
# rocprofv3 kernel statistics of the cfg-2 bench step (development / evidence tool).  usage: tools/prof_cfg2.sh TAG [bench.py arguments]
#   -> gpurun_out/TAG_kernel_stats.txt (per-kernel table per step) and gpurun_out/TAG_prof.log
TAG=$1; shift
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-alt --no-other "$@" > $R/gpurun_out/${TAG}_prof.log 2>&1
cd $R; f=$(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1)
python tools/summarize_kernel_stats.py $f 7 "$TAG: bench.py cfg2 --steps 5 --warmup 2 $* (7 steps traced)" > gpurun_out/${TAG}_kernel_stats.txt
t=$(find gpurun_out/${TAG}_prof -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/step_timeline.py $t 7 > gpurun_out/${TAG}_step_timeline.txt
find gpurun_out/${TAG}_prof -name "*kernel_trace.csv" -delete
