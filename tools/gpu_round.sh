mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02i_gpu_tests.log 2>&1; echo "pytest rc=$?" 
tail -5 gpurun_out/r02i_gpu_tests.log
python bench.py > gpurun_out/r02i_bench.json 2> gpurun_out/r02i_bench.err; echo "bench rc=$?"; cat gpurun_out/r02i_bench.json
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02i_prof -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-alt > $R/gpurun_out/r02i_prof.log 2>&1
cd $R; f=$(find gpurun_out/r02i_prof -name "*kernel_stats.csv" | head -1); python tools/summarize_kernel_stats.py $f 7 "r02i: bench.py cfg2 --steps 5 --warmup 2 (7 steps traced)" > gpurun_out/r02i_kernel_stats.txt; head -50 gpurun_out/r02i_kernel_stats.txt
find gpurun_out/r02i_prof -name "*kernel_trace.csv" -delete
