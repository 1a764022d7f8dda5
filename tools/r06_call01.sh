#!/bin/bash
# round 6, call 1: does the driver's command print a line it can parse (<= 6 KB)?  + the package probe of VERDICT item 8
mkdir -p gpurun_out
python - <<'P' > gpurun_out/r06_package_probe.log 2>&1
import importlib
for m in ("diffusers", "kornia", "torchvision", "clip", "open_clip"):
    try:
        importlib.import_module(m); print(m, "importable")
    except Exception as e:
        print(m, "missing:", type(e).__name__)
P
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --dump-launches gpurun_out/r06_launches_b1_base.csv > gpurun_out/r06_bench_b1_base.log 2> gpurun_out/r06_bench_b1_base.err ) 2>&1 | tail -4
tail -1 gpurun_out/r06_bench_b1_base.log | wc -c
tail -1 gpurun_out/r06_bench_b1_base.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['extra_configs'])"
tail -3 gpurun_out/r06_bench_b1_base.err
cat gpurun_out/r06_package_probe.log
