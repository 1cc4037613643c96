for v in 0 1 0 1; do FSC_CONV_STATS=$v python bench.py --no-cpu-baseline --no-alt 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); c=d['roofline']['conv_ms_per_step']; print('stats=$v', round(d['value'],1), round(d['ms_per_step'],3), 'fwd', round(c['conv_l16_fwd_kernel'],3), 'wgrad', round(c['conv_l16_wgrad_kernel'],3))"; done
R=$PWD; cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
FSC_CONV_STATS=$v timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_prof$v -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-alt > /dev/null 2>&1
f=$(find $R/gpurun_out/ab_prof$v -name "*kernel_stats.csv" | head -1); python $R/tools/summarize_kernel_stats.py $f 7 "stats=$v" > $R/gpurun_out/ab_stats$v.txt
find $R/gpurun_out/ab_prof$v -name "*kernel_trace.csv" -delete
done
