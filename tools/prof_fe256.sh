# Development: rocprofv3 kernel stats of tools/frontend256_bench.py (the n_fft = 256 front-end at cfg 3's shape) -> gpurun_out/r04_frontend256_stats.txt
R=$PWD; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fe256 -o p -- python $R/tools/frontend256_bench.py > $R/gpurun_out/fe256.log 2>&1
cd $R; f=$(find gpurun_out/fe256 -name "*kernel_stats.csv" | head -1)
(echo "# rocprofv3 --kernel-trace --stats -- python tools/frontend256_bench.py  (128 x 441 000 samples, stft_256_128, log magnitude; 22 launches)"; head -4 $f | cut -d, -f1-8; grep -h "stft_256_128" gpurun_out/fe256.log) > gpurun_out/r04_frontend256_stats.txt
find gpurun_out/fe256 -name "*kernel_trace.csv" -delete; find gpurun_out/fe256 -name "*.db" -delete
cat gpurun_out/r04_frontend256_stats.txt
