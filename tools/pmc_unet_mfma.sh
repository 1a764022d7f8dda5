# MFMA-busy and LDS / VALU activity of the kernels over eager UNet evaluations (two separate --pmc passes, no tracing):
#   bash tools/pmc_unet_mfma.sh   (repo root, GPU box)  -> gpurun_out/${DTP_ROUND:-r05}_pmc_unet_mfma_{1,2}.csv + ..._pmc_unet_mfma.json
R=${DTP_ROUND:-r05}
export DTP_TUNE_CACHE=${DTP_TUNE_CACHE:-/tmp/tc_pmc.txt}
timeout 300 python tools/pmc_unet.py > gpurun_out/pmc_unet_warm.log 2>&1
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  rm -rf /tmp/pmcm_$i
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmcm_$i -o p -- python /root/repo/tools/pmc_unet.py > /tmp/pmcm_$i.log 2>&1
  f=$(find /tmp/pmcm_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python /root/repo/tools/pmc_agg.py $f /root/repo/gpurun_out/${R}_pmc_unet_mfma_$i.csv; else tail -5 /tmp/pmcm_$i.log > /root/repo/gpurun_out/${R}_pmc_unet_mfma_$i.err; fi
done
cd /root/repo
python tools/pmc_mfma_json.py gpurun_out/${R}_pmc_unet_mfma_1.csv gpurun_out/${R}_pmc_unet_mfma_2.csv gpurun_out/${R}_pmc_unet_mfma.json
