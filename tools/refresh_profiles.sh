# Round-2 profile refresh, one gpurun call (run from the repo root on the GPU box): bench lines, rocprofv3 kernel stats, tune table.
export DTP_TUNE_CACHE=/tmp/tc.txt
timeout 1200 python bench.py > gpurun_out/r02_b1.log 2>gpurun_out/r02_b1.err
timeout 600 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r02_b8.log 2>gpurun_out/r02_b8.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r02 -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-extras > /root/repo/gpurun_out/r02_prof.log 2>&1
find /tmp/prof -name "*kernel_stats*" -exec cp {} /root/repo/gpurun_out/r02_kernel_stats.csv \;
cp /tmp/tc.txt /root/repo/gpurun_out/r02_tune_cache.txt
cd /root/repo
