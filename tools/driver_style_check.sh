#!/bin/bash
# what the driver does at round end: smoke(), then the default bench line (timed), with nothing but the shipped tune table
mkdir -p gpurun_out
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | grep -v amdgpu.ids | tail -6
( time timeout 900 python bench.py > gpurun_out/${DTP_ROUND:-r06}_driver_style_bench.log 2> gpurun_out/${DTP_ROUND:-r06}_driver_style_bench.err ) 2>&1 | tail -4
tail -1 gpurun_out/${DTP_ROUND:-r06}_driver_style_bench.log | cut -c1-260
tail -3 gpurun_out/${DTP_ROUND:-r06}_driver_style_bench.err
