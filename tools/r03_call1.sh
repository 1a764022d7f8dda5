# round-3 first call: package probe, the new multi-rank bench tests, baseline bench with the per-launch dump
mkdir -p gpurun_out
python - > gpurun_out/r03_package_probe.log 2>&1 <<'PY'
import importlib
for m in ("diffusers", "kornia", "torchvision", "clip", "tornado", "open_clip"):
    try:
        mod = importlib.import_module(m); print(m, "IMPORTABLE", getattr(mod, "__version__", "?"))
    except Exception as e:
        print(m, "missing:", type(e).__name__, e)
PY
pip download diffusers==0.12.0 -d /tmp/x --no-deps >> gpurun_out/r03_package_probe.log 2>&1 | tail -2
export DTP_TUNE_CACHE=/tmp/tc.txt
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q --durations=10 > gpurun_out/r03_dist_tests.log 2>&1
timeout 900 python bench.py --dump-launches gpurun_out/r03_launches_b1_base.csv > gpurun_out/r03_base_b1.log 2>gpurun_out/r03_base_b1.err
nproc >> gpurun_out/r03_package_probe.log; df -h /dev/shm >> gpurun_out/r03_package_probe.log
