R=$PWD; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
timeout -k 5 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl2 -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-alt --no-other > $R/gpurun_out/tl2.log 2>&1
cd $R; f=$(find gpurun_out/tl2 -name "*kernel_trace.csv" | head -1); python tools/step_timeline.py $f 5 > gpurun_out/r04_step_timeline.txt
find gpurun_out/tl2 -name "*.csv" -delete; find gpurun_out/tl2 -name "*.db" -delete
head -5 gpurun_out/r04_step_timeline.txt; tail -3 gpurun_out/tl2.log
