# Round-2 profile refresh, one gpurun call (run from the repo root on the GPU box).  Order matters: the PMC traffic summary is
# collected first and copied to profiles/ so that the bench lines of the same call can carry it (bench.py reports it only when its
# kernel-source hash matches the running build).  Everything lands in gpurun_out/r02_*; copy what should be judged into profiles/.
bash tools/pmc_unet.sh
cp gpurun_out/r02_pmc_unet_traffic.json profiles/r02_pmc_unet_traffic.json
export DTP_TUNE_CACHE=/tmp/tc.txt
timeout 1200 python bench.py > gpurun_out/r02_b1.log 2>gpurun_out/r02_b1.err
timeout 600 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r02_b8.log 2>gpurun_out/r02_b8.err
DTP_FP8=1 timeout 600 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r02_b8_fp8.log 2>gpurun_out/r02_b8_fp8.err
timeout 600 python bench.py --cpu-config0 > gpurun_out/r02_cpu_config0.json 2>gpurun_out/r02_cpu_config0.err
DTP_FULLSIZE=1 DTP_FULLSIZE_JSON=gpurun_out/r02_fullsize_parity.json timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k test_config1_512_20steps_matches_cpu_oracle > gpurun_out/r02_fullsize_parity.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r02 -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-extras > /root/repo/gpurun_out/r02_prof.log 2>&1
find /tmp/prof -name "*kernel_stats*" -exec cp {} /root/repo/gpurun_out/r02_kernel_stats.csv \;
rm -rf /tmp/prof8
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof8 -o r02 -- python /root/repo/bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-extras > /root/repo/gpurun_out/r02_prof_b8.log 2>&1
find /tmp/prof8 -name "*kernel_stats*" -exec cp {} /root/repo/gpurun_out/r02_kernel_stats_b8.csv \;
cp /tmp/tc.txt /root/repo/gpurun_out/r02_tune_cache.txt
cd /root/repo
