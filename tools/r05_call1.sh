#!/bin/bash
# round 5, call 1: package probe (verdict item 8), attention probes, baseline attention launch times
mkdir -p gpurun_out
{
for m in diffusers kornia torchvision clip tornado open_clip; do python -c "import $m; print('$m OK', getattr($m,'__version__','?'))" 2>&1 | tail -1; done
ls /opt/wheelhouse 2>/dev/null | grep -i -E "diffusers|kornia|torchvision" || echo "wheelhouse: none of diffusers/kornia/torchvision"
} > gpurun_out/r05_package_probe.log 2>&1
timeout 300 tools/micro/attn_probe > gpurun_out/r05_attn_probe.log 2>&1
timeout 600 python tools/bench_attn.py > gpurun_out/r05_bench_attn_head.log 2>&1
tail -5 gpurun_out/r05_package_probe.log; tail -40 gpurun_out/r05_attn_probe.log; cat gpurun_out/r05_bench_attn_head.log
