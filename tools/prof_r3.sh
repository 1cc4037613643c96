# rocprofv3 kernel stats of bench.py cfg2 (development tool): tools/prof_r3.sh <tag>
TAG=${1:-r03a}; R=$PWD; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-alt > $R/gpurun_out/${TAG}_prof.log 2>&1
cd $R; f=$(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1); python tools/summarize_kernel_stats.py $f 7 "${TAG}: bench.py cfg2 --steps 5 --warmup 2 (7 steps traced)" > gpurun_out/${TAG}_kernel_stats.txt
find gpurun_out/${TAG}_prof -name "*kernel_trace.csv" -delete; find gpurun_out/${TAG}_prof -name "*.db" -delete
