# The bench lines and rocprofv3 kernel statistics of tools/refresh_profiles_r04.sh alone (no PMC passes, no diagnostics): for a second
# box when the first refresh landed on a slow one.  Output: gpurun_out/r04_{b1,b8,256}.log, r04_kernel_stats{,_b8}.csv
export DTP_ROUND=r04
timeout 1200 python bench.py > gpurun_out/r04_b1.log 2>gpurun_out/r04_b1.err
timeout 600 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r04_b8.log 2>gpurun_out/r04_b8.err
timeout 600 python bench.py --res 256 --no-cpu-baseline --no-extras > gpurun_out/r04_256.log 2>gpurun_out/r04_256.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof /tmp/prof8
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r04 -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-extras > /root/repo/gpurun_out/r04_prof.log 2>&1
find /tmp/prof -name "*kernel_stats*" -exec cp {} /root/repo/gpurun_out/r04_kernel_stats.csv \;
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof8 -o r04 -- python /root/repo/bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-extras > /root/repo/gpurun_out/r04_prof_b8.log 2>&1
find /tmp/prof8 -name "*kernel_stats*" -exec cp {} /root/repo/gpurun_out/r04_kernel_stats_b8.csv \;
cd /root/repo
for f in b1 b8 256; do grep "^{" gpurun_out/r04_$f.log | tail -1 | grep -o "\"ms_per_step\": [0-9.]*"; done
