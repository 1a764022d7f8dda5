# Counter evidence for the GroupNorm-on-the-staged-patch fusion (DTP_GN_CONV=1) against the default build: SQ counters per kernel
# over three eager UNet evaluations (separate --pmc passes, no tracing).  -> gpurun_out/r03_pmc_gnconv_{off,on}_{1,2,3}.csv
cd /tmp && export TMPDIR=/tmp
for mode in off on; do
  if [ $mode = on ]; then export DTP_GN_CONV=1; else unset DTP_GN_CONV; fi
  export DTP_TUNE_CACHE=/tmp/tc_gnconv_$mode.txt
  timeout 400 python /root/repo/tools/pmc_unet.py > /root/repo/gpurun_out/pmc_gnconv_warm_$mode.log 2>&1
  i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA"; do
    i=$((i+1))
    rm -rf /tmp/pmcg_${mode}_$i
    timeout 600 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmcg_${mode}_$i -o p -- python /root/repo/tools/pmc_unet.py > /tmp/pmcg_${mode}_$i.log 2>&1
    f=$(find /tmp/pmcg_${mode}_$i -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python /root/repo/tools/pmc_agg.py $f /root/repo/gpurun_out/r03_pmc_gnconv_${mode}_$i.csv; else tail -5 /tmp/pmcg_${mode}_$i.log > /root/repo/gpurun_out/r03_pmc_gnconv_${mode}_$i.err; fi
  done
done
