# kernel trace of a graph-replayed batch-1 bench run -> busy / idle split and gap histogram of the last stamp (tools/gap_analysis.py)
mkdir -p gpurun_out
export DTP_TUNE_CACHE=/tmp/tc.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/proft
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/proft -o r03 -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-extras > /root/repo/gpurun_out/r03_trace_run.log 2>&1
f=$(find /tmp/proft -name "*kernel_trace.csv" | head -1)
python /root/repo/tools/gap_analysis.py $f 5389 > /root/repo/gpurun_out/r03_gap_analysis_b1.log 2>&1
