R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c5_prof -o p -- python $R/bench.py --workload cfg5 --no-cpu-baseline --no-kernel-timer --no-alt --no-other > $R/gpurun_out/c5_prof.log 2>&1
cd $R; f=$(find gpurun_out/c5_prof -name "*kernel_stats.csv" | head -1)
python tools/summarize_kernel_stats.py $f 1 "cfg5 inference pass, f16x6 (+ f16x3 and bf16x9 re-passes in the same trace)" > gpurun_out/c5_kernel_stats.txt
find gpurun_out/c5_prof -name "*kernel_trace.csv" -delete
head -40 gpurun_out/c5_kernel_stats.txt
