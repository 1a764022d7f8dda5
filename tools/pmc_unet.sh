# HBM traffic of the implicit-GEMM kernels over UNet evaluations: separate --pmc passes (FETCH_SIZE, WRITE_SIZE), no tracing.
# Run from the repo root on the GPU box: bash tools/pmc_unet.sh   -> gpurun_out/${DTP_ROUND:-r03}_pmc_unet_{FETCH_SIZE,WRITE_SIZE}.csv, ${DTP_ROUND:-r03}_pmc_unet_traffic.json
export DTP_TUNE_CACHE=/tmp/tc_pmc.txt
timeout 400 python tools/pmc_unet.py > gpurun_out/pmc_unet_warm.log 2>&1
cd /tmp && export TMPDIR=/tmp
for cnt in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcu_$cnt
  timeout 600 rocprofv3 --pmc $cnt --output-format csv -d /tmp/pmcu_$cnt -o p -- python /root/repo/tools/pmc_unet.py > /tmp/pmcu_$cnt.log 2>&1
  f=$(find /tmp/pmcu_$cnt -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python /root/repo/tools/pmc_agg.py $f /root/repo/gpurun_out/${DTP_ROUND:-r03}_pmc_unet_$cnt.csv; else tail -5 /tmp/pmcu_$cnt.log > /root/repo/gpurun_out/${DTP_ROUND:-r03}_pmc_unet_$cnt.err; fi
done
cd /root/repo
python tools/pmc_traffic_json.py gpurun_out/${DTP_ROUND:-r03}_pmc_unet_FETCH_SIZE.csv gpurun_out/${DTP_ROUND:-r03}_pmc_unet_WRITE_SIZE.csv gpurun_out/${DTP_ROUND:-r03}_pmc_unet_traffic.json
