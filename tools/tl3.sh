R=$PWD; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
timeout -k 5 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl3 -o p -- python $R/bench.py --workload cfg3 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-alt --no-other > $R/gpurun_out/tl3.log 2>&1
cd $R; f=$(find gpurun_out/tl3 -name "*kernel_trace.csv" | head -1); python tools/step_timeline.py $f 5 > gpurun_out/r04_step_timeline_cfg3.txt
find gpurun_out/tl3 -name "*.csv" -delete; find gpurun_out/tl3 -name "*.db" -delete
head -5 gpurun_out/r04_step_timeline_cfg3.txt; tail -3 gpurun_out/tl3.log
