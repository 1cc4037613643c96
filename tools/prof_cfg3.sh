R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cfg3_prof -o p -- python $R/bench.py --workload cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-alt --no-graph --no-other > $R/gpurun_out/cfg3_prof.log 2>&1
cd $R; f=$(find gpurun_out/cfg3_prof -name "*kernel_stats.csv" | head -1); python tools/summarize_kernel_stats.py $f 7 "cfg3 bf16" > gpurun_out/cfg3_kernel_stats.txt; head -40 gpurun_out/cfg3_kernel_stats.txt; tail -1 gpurun_out/cfg3_kernel_stats.txt
find gpurun_out/cfg3_prof -name "*kernel_trace.csv" -delete
