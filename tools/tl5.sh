# Development: rocprofv3 kernel stats of the cfg-5 inference workload (5-fold, length-grouped) -> gpurun_out/r04_kernel_stats_cfg5.txt
R=$PWD; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tl5 -o p -- python $R/bench.py --workload cfg5 --warmup 1 --no-alt --no-cpu-baseline --no-other --no-kernel-timer > $R/gpurun_out/tl5.log 2>&1
cd $R; f=$(find gpurun_out/tl5 -name "*kernel_stats.csv" | head -1); python tools/summarize_kernel_stats.py $f 1 "r04: bench.py --workload cfg5 (whole run: warm-up pass + timed pass over 56 batches x 5 folds)" > gpurun_out/r04_kernel_stats_cfg5.txt
find gpurun_out/tl5 -name "*kernel_trace.csv" -delete; find gpurun_out/tl5 -name "*.db" -delete
head -40 gpurun_out/r04_kernel_stats_cfg5.txt; tail -2 gpurun_out/r04_kernel_stats_cfg5.txt; tail -3 gpurun_out/tl5.log | cut -c1-300
