#!/bin/bash
# round 6: the fp8 Linear probe (verdict item 7), then the profile refresh
mkdir -p gpurun_out
timeout 600 tools/micro/fp8_linear_probe > gpurun_out/r06_fp8_linear_probe.log 2>&1; cat gpurun_out/r06_fp8_linear_probe.log
bash tools/refresh_profiles_r06.sh
