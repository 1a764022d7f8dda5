export DTP_TUNE_CACHE=/tmp/tc.txt
timeout 900 python bench.py > gpurun_out/r01_b1.log 2>&1
timeout 600 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r01_b8.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r01 -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > /root/repo/gpurun_out/r01_prof.log 2>&1
cp /tmp/prof/*kernel_stats.csv /root/repo/gpurun_out/r01_kernel_stats.csv 2>/dev/null || find /tmp/prof -name "*kernel_stats*" -exec cp {} /root/repo/gpurun_out/r01_kernel_stats.csv \;
cp /tmp/tc.txt /root/repo/gpurun_out/r01_tune_cache.txt
