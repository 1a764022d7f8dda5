# Round-6 profile refresh, one gpurun call (run from the repo root on the GPU box).  Order matters: the PMC summaries (HBM traffic,
# MFMA-busy) are collected first and copied to profiles/ so that the bench lines of the same call can carry them (bench.py reports them
# only when their kernel-source hash matches the running build).  Everything lands in gpurun_out/r06_*; copy what should be judged into profiles/.
export DTP_ROUND=r06
export DTP_TUNE_CACHE=/tmp/tc.txt
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/tc.txt   # shapes the shipped table lacks are tuned here and appended: the result is the next shipped table
python - > gpurun_out/r06_package_probe.log 2>&1 <<'PY'
import importlib
for m in ("diffusers", "kornia", "torchvision", "clip", "tornado", "transformers", "onnx", "tensorrt"):
    try:
        importlib.import_module(m); print(m, "importable")
    except Exception as e:
        print(m, "missing:", type(e).__name__)
PY
bash tools/pmc_unet.sh
cp gpurun_out/r06_pmc_unet_traffic.json profiles/r06_pmc_unet_traffic.json
bash tools/pmc_unet_mfma.sh > gpurun_out/r06_pmc_unet_mfma.log 2>&1
cp gpurun_out/r06_pmc_unet_mfma.json profiles/r06_pmc_unet_mfma.json
timeout 1500 python bench.py --dump-launches gpurun_out/r06_launches_b1.csv > gpurun_out/r06_b1.log 2>gpurun_out/r06_b1.err
timeout 600 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r06_b8.log 2>gpurun_out/r06_b8.err
timeout 600 python bench.py --res 256 --no-cpu-baseline --no-extras > gpurun_out/r06_256.log 2>gpurun_out/r06_256.err
DTP_BENCH_BACKEND=gloo DTP_BENCH_SAME_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r06_gpus2_same_device.log 2>gpurun_out/r06_gpus2_same_device.err
DTP_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r06_rccl_one_rank.log 2>gpurun_out/r06_rccl_one_rank.err
if [ -z "$DTP_REFRESH_SKIP_ORACLE" ]; then  # (the two long CPU-oracle legs: configs[0] in full on the host, configs[1] + configs[2] at full size -- the GPU suite below repeats the latter)
timeout 600 python bench.py --cpu-config0 > gpurun_out/r06_cpu_config0.json 2>gpurun_out/r06_cpu_config0.err
DTP_FULLSIZE_JSON=gpurun_out/r06_fullsize_parity.json timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k test_config1_and_config2_512_20steps_match_cpu_oracle > gpurun_out/r06_fullsize_parity.log 2>&1
fi
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r06 -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-extras > /root/repo/gpurun_out/r06_prof.log 2>&1
find /tmp/prof -name "*kernel_stats*" -exec cp {} /root/repo/gpurun_out/r06_kernel_stats.csv \;
rm -rf /tmp/prof8
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof8 -o r06 -- python /root/repo/bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-extras > /root/repo/gpurun_out/r06_prof_b8.log 2>&1
find /tmp/prof8 -name "*kernel_stats*" -exec cp {} /root/repo/gpurun_out/r06_kernel_stats_b8.csv \;
if [ -f /tmp/tc.txt ]; then cp /tmp/tc.txt /root/repo/gpurun_out/r06_tune_cache.txt; else cp /root/repo/diffusiontexturepainting_amd/tune_seed.txt /root/repo/gpurun_out/r06_tune_cache.txt; fi
cd /tmp && rm -rf /tmp/profm
timeout 600 rocprofv3 --marker-trace --stats --output-format csv -d /tmp/profm -o r06 -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-extras > /root/repo/gpurun_out/r06_prof_marker.log 2>&1
find /tmp/profm -name "*marker*stats*" -exec cp {} /root/repo/gpurun_out/r06_marker_stats.csv \;
cd /root/repo
if [ -n "$DTP_REFRESH_AB" ]; then  # same-box A/B against a reference library (e.g. tools/ab/libdtp_r05.so = round 5 rebuilt from git), shipped tune table on both arms
  rm -f gpurun_out/ab_summary.log
  cp diffusiontexturepainting_amd/tune_seed.txt /tmp/ab_tc.txt
  DTP_TUNE_CACHE=/tmp/ab_tc.txt bash tools/ab.sh $DTP_REFRESH_AB all 2 > /dev/null 2>&1
  cp gpurun_out/ab_summary.log gpurun_out/r06_ab_r05_vs_r06_final.txt
fi
( time DTP_FULLSIZE_JSON=gpurun_out/r06_fullsize_parity.json timeout 1500 python -m pytest tests -q -m gpu --durations=15 -x ) > gpurun_out/r06_gpu_suite.log 2>&1
tail -3 gpurun_out/r06_gpu_suite.log
cp /tmp/tc.txt gpurun_out/r06_tune_cache.txt
for f in b1 b8 256; do grep "^{" gpurun_out/r06_$f.log | tail -1 | grep -o "\"ms_per_step\": [0-9.]*"; done
