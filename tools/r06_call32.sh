#!/bin/bash
# round 6, last commit: smoke, the tests around the runtime switch (default = captured packets), the default bench line
mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 900 python -m pytest tests/test_gpu_round2.py -q -m gpu -x -k "eager or enqueue or handler or tune_table" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_dist.py -q -m gpu -x 2>&1 | tail -3
( time timeout 900 python bench.py > gpurun_out/r06_last_bench.log 2>/dev/null ) 2>&1 | tail -3
tail -1 gpurun_out/r06_last_bench.log | cut -c1-300
