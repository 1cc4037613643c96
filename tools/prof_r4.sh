# Round-4 profile set (run on the GPU box): rocprofv3 kernel stats of the cfg-2 and cfg-3 bench steps, HBM traffic of the conv kernels
# (PMC FETCH_SIZE / WRITE_SIZE in their own passes) -> gpurun_out/r04_*; copy what is kept into profiles/.   tools/prof_r4.sh <tag>
TAG=${1:-r04}; R=$PWD; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-alt --no-other > $R/gpurun_out/${TAG}_prof.log 2>&1
cd $R; f=$(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1); python tools/summarize_kernel_stats.py $f 7 "${TAG}: bench.py cfg2 --steps 5 --warmup 2 (7 steps traced)" > gpurun_out/${TAG}_kernel_stats.txt
find gpurun_out/${TAG}_prof -name "*kernel_trace.csv" -delete; find gpurun_out/${TAG}_prof -name "*.db" -delete
cd /tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof3 -o p -- python $R/bench.py --workload cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-alt --no-graph --no-other > $R/gpurun_out/${TAG}_prof3.log 2>&1
cd $R; f=$(find gpurun_out/${TAG}_prof3 -name "*kernel_stats.csv" | head -1); python tools/summarize_kernel_stats.py $f 7 "${TAG}: bench.py --workload cfg3 (bf16, eager: --no-graph) --steps 5 --warmup 2 (7 steps traced)" > gpurun_out/${TAG}_kernel_stats_bench_cfg3.txt
find gpurun_out/${TAG}_prof3 -name "*kernel_trace.csv" -delete; find gpurun_out/${TAG}_prof3 -name "*.db" -delete
bash tools/collect_traffic.sh > gpurun_out/${TAG}_traffic.log 2>&1
head -30 gpurun_out/${TAG}_kernel_stats.txt; tail -1 gpurun_out/${TAG}_kernel_stats.txt; tail -1 gpurun_out/${TAG}_kernel_stats_bench_cfg3.txt; tail -5 gpurun_out/${TAG}_traffic.log
